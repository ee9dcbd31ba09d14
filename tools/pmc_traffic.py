#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE PMC passes of bench.py (tools/gpu_round3.sh) into
profiles/<tag>_pmc_traffic.json and profiles/pmc_traffic.json (read by bench.py for roofline.traffic).
Usage: python3 tools/pmc_traffic.py gpurun_out/<dir> <tag>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, T = 256, 199
M = B * T
L = [12799, 6399, 3199, 1599, 799, 399, 199]


def algorithmic_by_kernel():
    """Algorithmic bytes (A + W + C + residual) per launch, split by the instantiation the single-stream forward runs at
    B = 256 x 4 s (round 4: every conv / proj / transformer GEMM takes 256x128 tiles, alone or in a two-shape launch with 128x128
    tiles for the rows of the last round; key "gemm_128x64" = what is left on finer kernels: the pos-conv's N = 48 kernel)."""
    big, fine = [], []
    for i in range(1, 7):
        k = 3 if i < 5 else 2
        big.append(B * L[i - 1] * 512 * 4 + 512 * 512 * k * 4 + B * L[i] * 512 * 4)
    big.append(M * 512 * 4 + 768 * 512 * 4 + M * 768 * 4)
    fine.append(B * (T + 128) * 768 * 4 * 2 + 16 * 64 * 6144 * 4 + M * 768 * 4)
    for _ in range(12):
        big.append(M * 768 * 4 + 2304 * 768 * 4 + M * 2304 * 4)
        big.append(M * 768 * 4 + 768 * 768 * 4 + 2 * M * 768 * 4)
        big.append(M * 768 * 4 + 3072 * 768 * 4 + M * 3072 * 4)
        big.append(M * 3072 * 4 + 3072 * 768 * 4 + 2 * M * 768 * 4)
    return sum(big) / len(big), sum(fine) / len(fine)


def algorithmic_gemm_bytes():
    """A + W + C (+ residual) of every GEMM launch of one forward at B = 256 x 4 s, fp32."""
    alg, n = 0, 0
    for i in range(1, 7):
        k = 3 if i < 5 else 2
        alg += B * L[i - 1] * 512 * 4 + 512 * 512 * k * 4 + B * L[i] * 512 * 4
        n += 1
    alg += M * 512 * 4 + 768 * 512 * 4 + M * 768 * 4
    n += 1                                                       # post_extract_proj
    alg += B * (T + 128) * 768 * 4 * 2 + 16 * 64 * 6144 * 4 + M * 768 * 4
    n += 1                                                       # pos-conv (A + residual + W + C)
    for _ in range(12):
        alg += M * 768 * 4 + 2304 * 768 * 4 + M * 2304 * 4       # qkv
        alg += M * 768 * 4 + 768 * 768 * 4 + 2 * M * 768 * 4     # out_proj (+ residual)
        alg += M * 768 * 4 + 3072 * 768 * 4 + M * 3072 * 4       # fc1
        alg += M * 3072 * 4 + 3072 * 768 * 4 + 2 * M * 768 * 4   # fc2 (+ residual)
        n += 4
    return alg / n, n


def summarise(src, tag):
    """src: directory holding pmc_FETCH_SIZE/ and pmc_WRITE_SIZE/ (rocprofv3 -d outputs of the two passes) -> the summary dict
    (bench.py's roofline.traffic reads "gemm_256x128" / "gemm_128x64" of it)."""
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(src, "pmc_%s" % c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "nomad" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    name = r["Kernel_Name"].replace("void nomad::", "").replace("(nomad::GemmParams)", "").split("(")[0]
                    agg[name + " grid=" + r["Grid_Size"]][c].append(float(r["Counter_Value"]) * 1024.0)
    alg, nl = algorithmic_gemm_bytes()
    out = {
        "taken": tag,
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py "
                  "--steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-also --single-stream; MI355X (tools/gpu_pmc_traffic.sh)",
        "units": "bytes per launch.  The counters sit on the fabric side of L2 and count Infinity-Cache hits: an upper "
                 "bound on HBM bytes.",
        "calibration": "gfx950 FETCH_SIZE halves wide coalesced streaming reads (MI355X_MICROARCH.md, HBM): confirmed on "
                       "layernorm_kernel (16 B/lane, 1 KiB per wave-instruction).  The LDS-DMA GEMM loads fetch 64-B / 128-B "
                       "row segments and are counted exactly (conv1: raw fetch ~= algorithmic A + W).  fetch_factor 1 for "
                       "gemm kernels, 2 for the row-streaming kernels.",
        "kernels": {},
    }
    gl = gf = gw = 0
    for k, v in sorted(agg.items()):
        f, w = v.get("FETCH_SIZE", []), v.get("WRITE_SIZE", [])
        fac = 1 if "gemm" in k else 2
        out["kernels"][k] = {"launches": len(f), "fetch_factor": fac,
                             "fetch_bytes_raw_per_launch": sum(f) / max(len(f), 1),
                             "fetch_bytes_per_launch": fac * sum(f) / max(len(f), 1),
                             "write_bytes_per_launch": sum(w) / max(len(w), 1)}
        if "gemm" in k:
            gl += len(f)
            gf += sum(f)
            gw += sum(w)
    out["gemm_all_launches"] = {"launches": gl, "hbm_bytes_per_launch": (gf + gw) / gl, "fetch": gf / gl, "write": gw / gl,
                                "algorithmic_bytes_per_launch": alg, "ratio": (gf + gw) / gl / alg}
    # "gemm_128x64" keeps its name (bench.py's key) but now covers the finer instantiations: 128x128x32 and 128x64x32
    for key, pats in (("gemm_256x128", ("<256, 128,", "gemm_f32_mixed_kernel")), ("gemm_128x64", ("<128, 64,", "<128, 128,", "n48_kernel"))):
        sel = [v for k, v in out["kernels"].items() if "gemm" in k and any(p in k for p in pats)]
        n = sum(v["launches"] for v in sel)
        b = sum(v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"]) for v in sel)
        out[key] = {"launches": n, "hbm_bytes_per_launch": b / max(n, 1),
                    "algorithmic_bytes_per_launch": algorithmic_by_kernel()[0 if key == "gemm_256x128" else 1]}
    return out


def collect(out_dir, timeout=180):
    """Run the two PMC passes of the bench command as child processes (rocprofv3 --kernel-trace --pmc C -- python3 bench.py ...:
    separate passes, no other tracing) into out_dir and summarise them.  Called by bench.py itself on a 1-GPU headline run, so
    that roofline.traffic is measured in the same run; raises on any failure (the caller falls back to profiles/pmc_traffic.json)."""
    import shutil
    import subprocess
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        d = os.path.join(out_dir, "pmc_%s" % c)
        os.makedirs(d, exist_ok=True)
        cmd = [exe, "--kernel-trace", "--pmc", c, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-profile", "--no-also",
               "--single-stream", "--live-traffic", "off"]
        with open(os.path.join(out_dir, "pmc_%s.log" % c), "w") as log:
            rc = subprocess.call(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=timeout)
        if rc != 0:
            raise RuntimeError("rocprofv3 --pmc %s pass exited with %d" % (c, rc))
    out = summarise(out_dir, "live")
    if not out["gemm_256x128"]["launches"]:
        raise RuntimeError("no GEMM launches in the counter files")
    return out


if __name__ == "__main__":
    src, tag = sys.argv[1], sys.argv[2]
    out = summarise(src, tag)
    # also next to the passes: gpurun_out/ is what travels back from the GPU box, profiles/ on the box does not
    for dst in (os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), os.path.join(ROOT, "profiles", "pmc_traffic.json"),
                os.path.join(src, "pmc_traffic_summary.json")):
        json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["gemm_all_launches"]))
    for k, v in out["kernels"].items():
        print(f"{k:60s} n={v['launches']:3d} fetch={v['fetch_bytes_per_launch'] / 1e6:9.1f} MB write={v['write_bytes_per_launch'] / 1e6:9.1f} MB")
