"""CPU: the C-ABI library builds, loads, and exports exactly what include/nomad_hip.h declares.
No compute entry point is called here (there is no GPU in the build container)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "nomad_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nomad_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(built_lib):
    from nomad_amd import _lib
    names = header_functions()
    assert len(names) >= 15
    assert sorted(_lib.SIGNATURES) == names
    lib = C.CDLL(built_lib)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/nomad_hip.h but not exported"


def test_shape_helpers(built_lib):
    from nomad_amd import _lib
    from nomad_amd.weights import num_frames
    lib = _lib.load()
    assert lib.nomad_version().startswith(b"nomad_hip 0.3")
    # binding, header and library agree on the binary-interface number (bumped when nomad_set_concurrent_parts gained its context argument)
    hdr = open(os.path.join(ROOT, "include", "nomad_hip.h")).read()
    assert int(re.search(r"#define NOMAD_ABI_VERSION (\d+)", hdr).group(1)) == _lib.ABI_VERSION == lib.nomad_abi_version() == 3
    for n in (9, 400, 16384, 27225, 64000, 223840, 480000):
        assert lib.nomad_num_frames(n) == max(num_frames(n), 0)
    assert lib.nomad_l1_scratch_bytes() > 0
    # the shipped library reports a build without packed-FP32 instructions and without the diagnostic instantiations; the
    # diagnostic one says so (nomad_build_flags: Engine keys its two-stream splits on bit 0)
    assert lib.nomad_build_flags() == 0
    assert _lib.load(diag=True).nomad_build_flags() == 2


def test_product_library_reads_no_environment(built_lib):
    """include/nomad_hip.h: "nothing but the arguments" decides what a call does.  The kernel-choice switches of the A/B runs
    (NOMAD_F32_*, NOMAD_BF16_*, NOMAD_SPLITK_*) live in a per-context struct whose defaults are the shipped configuration; only
    libnomad_diag.so fills it from the environment.  Hold the product to that: it does not even import getenv."""
    import shutil
    import subprocess
    from nomad_amd import build
    nm = shutil.which("nm") or shutil.which("llvm-nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    if not os.path.exists(nm):
        pytest.skip("nm not available")
    syms = subprocess.run([nm, "-D", "--undefined-only", build.LIB], capture_output=True, text=True, check=True).stdout
    assert "hipLaunchKernel" in syms or "hipModuleLaunchKernel" in syms or "__hipPushCallConfiguration" in syms, "nm output looks empty"
    assert not re.search(r"\b(secure_)?getenv\b", syms), "libnomad_hip.so imports getenv"
    diag = subprocess.run([nm, "-D", "--undefined-only", build.DIAG_LIB], capture_output=True, text=True, check=True).stdout
    assert re.search(r"\bgetenv\b", diag), "libnomad_diag.so is expected to read its A/B switches from the environment"


def test_create_fails_loudly_without_gpu(built_lib, sd0):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nomad_amd.engine import Engine
    from nomad_amd._lib import NomadHipError
    with pytest.raises(NomadHipError, match="no HIP device|no CPU path|device"):
        Engine(sd0, 0)


def test_cpu_device_is_rejected():
    from nomad_amd.nomad import Nomad
    with pytest.raises(RuntimeError, match="no CPU path"):
        Nomad(device="cpu", weights="seeded")


def test_default_device_without_a_gpu_is_rejected_like_cpu():
    """Nomad(device=None): the reference falls back to the CPU (nomad.py:40-43); this build says there is no CPU path, at construction."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from nomad_amd.nomad import Nomad
    with pytest.raises(RuntimeError, match="no CPU path"):
        Nomad(weights="seeded")


def test_no_oracle_import_in_product():
    """The product must not route through the oracle."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nomad_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_shipped_libraries_have_no_packed_fp32_instructions(built_lib):
    """DESIGN.md "The packed-FP32 hazard": v_pk_fma_f32 & co lose products when a bf16 MFMA GEMM of another stream shares
    the SIMD, so nomad_amd/build.py compiles the device code without them.  Disassemble the gfx950 code objects inside the
    built libraries and hold it to that (the A/B builds *_pk.so are the ones WITH them)."""
    import re
    import shutil
    import struct
    import subprocess
    import tempfile
    objdump = shutil.which("llvm-objdump") or "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    from nomad_amd import build
    for lib in (build.LIB, build.DIAG_LIB):
        data = open(lib, "rb").read()
        starts = [m.start() for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data)]   # one bundle per translation unit (three since round 6)
        assert len(starts) >= 3, (lib, len(starts))
        found, attn_checked = 0, 0
        for i in starts:
            n = struct.unpack_from("<Q", data, i + 24)[0]
            off = i + 32
            for _ in range(n):
                o, sz, tl = struct.unpack_from("<QQQ", data, off)
                off += 24
                triple = data[off:off + tl].decode()
                off += tl
                if "gfx950" not in triple:
                    continue
                with tempfile.NamedTemporaryFile(suffix=".co") as f:
                    f.write(data[i + o:i + o + sz])
                    f.flush()
                    asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
                found += 1
                assert asm.count("s_endpgm") > 10, "disassembly looks empty"
                assert not re.search(r"v_pk_(fma|mul|add)_f32|v_pk_mov_b32", asm), f"{os.path.basename(lib)} contains packed-FP32 instructions"
                # Round 4, tools/micro/store_hazard.hip: a buffer store of more than 8 bytes whose soffset is an SGPR gets NO wait
                # state from the compiler before a VALU write of its data registers, and on gfx950 that corrupts lanes 12-15 / 28-31 /
                # 44-47 / 60-63 of the first data register when MFMA waves share the SIMD.  The kernels keep the row offset in the VGPR
                # offset instead; hold the libraries to "no wide buffer store with a register soffset".
                bad = re.findall(r"buffer_store_dwordx[34] v\[\d+:\d+\], v\d+, s\[\d+:\d+\], s\d+ offen", asm)
                assert not bad, f"{os.path.basename(lib)}: wide buffer stores with an SGPR soffset: {bad[:3]}"
                # Round 5 ("LDS reads hipcc does not order behind the next tile's LDS-DMA"): an LDS read without alias
                # information - a float4 STRUCT copy, the ds_read_tr builtin - gets s_waitcnt vmcnt(0) in front of it while an LDS-DMA is in
                # flight, which serialises a double-buffered kernel's prefetch with its products.  The attention kernels the forwards
                # launch read LDS through ext-vector loads / inline asm instead: hold them to "no vmcnt wait directly in front of a ds_read".
                if lib == build.LIB:
                    for name in ("attention_f32_v2_kernel", "attention_bf16_v3_kernel"):
                        bodies = re.findall(r"<_ZN5nomad\d+%s\w*>:\n(.*?)s_endpgm" % name, asm, re.S)
                        attn_checked += len(bodies)
                        for body in bodies:
                            assert "global_load_lds" in body, name
                            hits = re.findall(r"s_waitcnt vmcnt\(\d+\)[^\n]*\n\s*ds_read", body)
                            assert not hits, f"{name}: {len(hits)} vmcnt waits directly in front of LDS reads"
        assert found == len(starts), (lib, found, len(starts))
        assert lib != build.LIB or attn_checked >= 2, attn_checked
