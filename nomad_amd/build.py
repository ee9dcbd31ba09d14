"""Build the in-tree HIP libraries with hipcc for gfx950 (cross-compiles without a GPU).

* ``libnomad_hip.so``  - the product: only the kernel instantiations the scoring / training paths can select.
* ``libnomad_diag.so`` - the same source with ``-DNOMAD_DIAG``: additionally every experimental GEMM instantiation,
  ablation and timing probe (tools/gemm_sweep.py, tools/gemm_x3.py, the tests of those tiles).  Never loaded by the
  product path (``nomad_amd._lib.load()``); ``Engine(..., diag=True)`` / ``NOMAD_DIAG_LIB=1`` select it explicitly.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libnomad_hip.so")
DIAG_LIB = os.path.join(HERE, "libnomad_diag.so")
SOURCES = ["nomad_hip.hip", "nomad_gemm_f32.hip", "nomad_gemm_bf16.hip"]   # three translation units, compiled side by side (csrc/nomad_ctx.hip.h)
# every header the translation units include: *.hip.h kernels AND plain *.h host code (wav_reader.h)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "nomad_hip.h")]


def hipcc_path() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found; the HIP libraries cannot be built")


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


# Device code is generated without the packed-FP32 VALU instructions (DESIGN.md "The packed-FP32 hazard": a v_pk_fma_f32 can
# lose a product while a bf16 MFMA kernel of another stream shares its SIMD).  The flag reaches the host pass too, which
# prints "not a recognized feature for this target (ignoring feature)" - harmless, and swallowed with the rest of hipcc's
# stderr on success.  NOMAD_PACKED_FP32=1 builds WITH those instructions (A/B measurements, the reproducer of the hazard).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]


def pk_path(lib: str) -> str:
    """Where the A/B build WITH packed FP32 of a library goes (tools/race_hunt_*.py, NOMAD_LIB_VARIANT=pk)."""
    return lib[:-3] + "_pk.so"


# A/B builds of one source-level choice, as <library>_<name>.so (NOMAD_LIB_VARIANT=<name> loads them: measurement tools only).
VARIANTS = {"gelu1": ["-DNOMAD_GELU_BF16_FORM=1"],   # the bf16 epilogues' GELU (gemm_f32.hip.h gelu_bf16out) in its first form (sigmoid, 9 instructions)
            "gelu2": ["-DNOMAD_GELU_BF16_FORM=2"],   # ... and with the quartic tail (8 instructions, 6.2e-6)
            "gelu_tail6": ["-DNOMAD_GELU_F32_FORM=2"]}  # the fp32 / bf16x3 paths' GELU (gelu_erf) as a sextic tail (10 instructions, 2.8e-7: measured, not shipped)


def variant_path(lib: str, name: str) -> str:
    return lib[:-3] + f"_{name}.so"


def _flags(lib: str, diag: bool, verbose: bool):
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-pthread"]
    for name, defs in VARIANTS.items():
        if lib.endswith(f"_{name}.so"):
            flags = defs + flags
    if os.environ.get("NOMAD_PACKED_FP32", "0") != "1" and not lib.endswith("_pk.so"):
        flags = NO_PACKED_FP32 + flags
    else:
        flags.insert(0, "-DNOMAD_PACKED_FP32_BUILD=1")   # nomad_build_flags() reports it; Engine then keeps its two-stream split off
    if diag:
        flags.insert(0, "-DNOMAD_DIAG")
    if verbose:
        flags.insert(0, "-Rpass-analysis=kernel-resource-usage")
    return flags


class _Job:
    """One library: its translation units are compiled to objects in parallel, then linked."""

    def __init__(self, lib: str, diag: bool, verbose: bool = False):
        self.lib, self.t0 = lib, time.perf_counter()
        self.tmp = lib + f".tmp{os.getpid()}"
        self.flags = _flags(lib, diag, verbose)
        self.objs = [f"{lib}.{os.path.splitext(s)[0]}.{os.getpid()}.o" for s in SOURCES]
        self.procs = [subprocess.Popen([hipcc_path(), *self.flags, "-c", os.path.join(CSRC, s), "-o", o],
                                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for s, o in zip(SOURCES, self.objs)]

    def _cleanup(self):
        for f in self.objs + [self.tmp]:
            if os.path.exists(f):
                os.remove(f)

    def finish(self) -> float:
        errs = []
        for p in self.procs:
            out, err = p.communicate()
            if p.returncode != 0:
                errs.append(out + err)
        if errs:
            self._cleanup()
            raise RuntimeError("hipcc failed:\n" + "\n".join(errs))
        link = subprocess.run([hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", self.tmp, *self.objs],
                              capture_output=True, text=True)
        if link.returncode != 0:
            self._cleanup()
            raise RuntimeError("hipcc (link) failed:\n" + link.stdout + link.stderr)
        os.replace(self.tmp, self.lib)  # atomic: a concurrent dlopen never sees a half-written library
        self._cleanup()
        return round(time.perf_counter() - self.t0, 1)


def build_library(force: bool = False, verbose: bool = False, diag: bool = False) -> str:
    """(Re)build one library if its sources changed; returns its path."""
    lib = DIAG_LIB if diag else LIB
    if not force and not needs_build(lib):
        return lib
    _Job(lib, diag, verbose).finish()
    return lib


def build_all(force: bool = False) -> dict:
    """Both libraries, all six hipcc compilations side by side; -> {name: seconds} of what was rebuilt."""
    jobs = [_Job(lib, diag) for lib, diag in ((LIB, False), (DIAG_LIB, True)) if force or needs_build(lib)]
    return {os.path.basename(j.lib): j.finish() for j in jobs}


def build_pk_variants() -> dict:
    """libnomad_hip_pk.so / libnomad_diag_pk.so: the same libraries WITH the packed-FP32 instructions, for A/B runs."""
    jobs = [_Job(lib, diag) for lib, diag in ((pk_path(LIB), False), (pk_path(DIAG_LIB), True))]
    return {os.path.basename(j.lib): j.finish() for j in jobs}


def build_variant(name: str) -> dict:
    """libnomad_hip_<name>.so / libnomad_diag_<name>.so for one entry of VARIANTS."""
    jobs = [_Job(variant_path(lib, name), diag) for lib, diag in ((LIB, False), (DIAG_LIB, True))]
    return {os.path.basename(j.lib): j.finish() for j in jobs}


if __name__ == "__main__":
    import sys
    if "--pk" in sys.argv:
        print(build_pk_variants())
    elif "--variant" in sys.argv:
        print(build_variant(sys.argv[sys.argv.index("--variant") + 1]))
    elif "-v" in sys.argv:
        print(build_library(force=True, verbose=True, diag="--diag" in sys.argv))
    else:
        print(build_all(force=True))
