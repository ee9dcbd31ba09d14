#!/usr/bin/env python3
"""Round 5: per-shape kernel times INSIDE the config-C5 forward, from a rocprofv3 --kernel-trace CSV of
`bench.py --dtype bf16 --seconds 30 --batch 32 --single-stream` (one stream: every kernel alone on the GPU, in program order).
A transformer layer is the launch pattern  GEMM(qkv) attention GEMM(out) [GEMM tail] LN GEMM(fc1) GEMM(fc2) [GEMM tail] LN; the
script finds the attention launches and labels the GEMMs around them.
Usage: python tools/c5_layer_table.py <kernel_trace.csv> [--M 47968]"""
import argparse, collections, csv, json, statistics, sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--M", type=int, default=47968)
    a = ap.parse_args()
    rows = []
    with open(a.csv) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    names = [r[2] for r in rows]
    dur = [(r[1] - r[0]) / 1e3 for r in rows]   # us
    is_gemm = ["gemm_bf16" in n for n in names]
    is_attn = ["attention_bf16" in n for n in names]
    is_ln = ["layernorm_kernel" in n for n in names]
    acc = collections.defaultdict(list)
    for i, at in enumerate(is_attn):
        if not at:
            continue
        acc["attention"].append(dur[i])
        # backwards: the GEMM right before is qkv
        j = i - 1
        while j >= 0 and not is_gemm[j]:
            j -= 1
        if j >= 0:
            acc["qkv"].append(dur[j])
        # forwards: GEMMs until the first LN = out_proj (+ tail), then GEMMs until the second LN = fc1, fc2 (+ tail)
        k = i + 1
        g1 = []
        while k < len(rows) and not is_ln[k]:
            if is_gemm[k]:
                g1.append(dur[k])
            k += 1
        if g1:
            acc["out_proj"].append(sum(g1))
        if k < len(rows):
            acc["layernorm"].append(dur[k])
        k += 1
        g2 = []
        while k < len(rows) and not is_ln[k]:
            if is_gemm[k]:
                g2.append(dur[k])
            k += 1
        if g2:
            acc["fc1"].append(g2[0])
            acc["fc2"].append(sum(g2[1:]))
    M = a.M
    flops = {"qkv": 2.0 * M * 2304 * 768, "out_proj": 2.0 * M * 768 * 768, "fc1": 2.0 * M * 3072 * 768, "fc2": 2.0 * M * 768 * 3072,
             "attention": 4.0 * (M / 1499) * 12 * 1499 * 1499 * 64}
    out = {}
    for k, v in acc.items():
        med = statistics.median(v)
        out[k] = {"launches": len(v), "median_us": round(med, 1), "mean_us": round(statistics.mean(v), 1)}
        if k in flops:
            out[k]["tflops_median"] = round(flops[k] / med / 1e6, 1)
    # the conv stack and everything else, per forward pass
    nfw = max(1, len(acc["attention"]) // 12)
    other = collections.defaultdict(float)
    for n, d in zip(names, dur):
        key = n.split("(")[0].replace("void nomad::", "").replace("nomad::", "")[:60]
        other[key] += d
    out["per_pass_ms_by_kernel"] = {k: round(v / nfw / 1e3, 3) for k, v in sorted(other.items(), key=lambda kv: -kv[1])[:12]}
    out["passes"] = nfw
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
