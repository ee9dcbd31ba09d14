#!/usr/bin/env python3
"""configs[3] step from a `rocprofv3 --kernel-trace` CSV of tools/c4_profile.py: the forward (up to the L1 loss kernels) and the backward
(from l1_bwd_kernel on) separately - GPU span, sum of kernel durations, idle gaps, and the kernels by total time with their grids.
Usage: python3 tools/c4_trace_table.py <kernel_trace.csv> [steps_to_skip]"""
import csv, json, re, sys, collections

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
# steps are delimited by l1_final_kernel (one per forward)
idx = [i for i, r in enumerate(rows) if "l1_final_kernel" in r["Kernel_Name"]]
steps = []
for a, b in zip(idx[skip:-1], idx[skip + 1:]):
    # a step's backward starts at the first l1_bwd_kernel after l1_final a; its forward is what precedes l1_final b back to the previous backward's end
    seg = rows[a:b]
    bw0 = next((k for k, r in enumerate(seg) if "l1_bwd_kernel" in r["Kernel_Name"]), None)
    if bw0 is None:
        continue
    # backward ends at the conv0_bwd / last backward kernel: the next forward begins with wav_stats_kernel
    fw0 = next((k for k, r in enumerate(seg) if k > bw0 and "wav_stats_kernel" in r["Kernel_Name"]), len(seg))
    steps.append((seg[bw0:fw0], seg[fw0:]))


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    return n.replace("nomad::", "")[:70]


def table(segs, title):
    agg = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
    span = busy = 0.0
    for seg in segs:
        if not seg:
            continue
        s0 = min(int(r["Start_Timestamp"]) for r in seg)
        s1 = max(int(r["End_Timestamp"]) for r in seg)
        span += (s1 - s0) / 1e3
        # union of busy intervals
        iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
        cur0, cur1 = iv[0]
        for a, b in iv[1:]:
            if a > cur1:
                busy += (cur1 - cur0) / 1e3
                cur0, cur1 = a, b
            else:
                cur1 = max(cur1, b)
        busy += (cur1 - cur0) / 1e3
        for r in seg:
            k = agg[short(r["Kernel_Name"])]
            k[0] += 1
            k[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            k[2][(r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])] += 1
    n = max(1, len(segs))
    out = {"part": title, "steps": len(segs), "gpu_span_us_per_step": round(span / n, 1), "gpu_busy_us_per_step": round(busy / n, 1),
           "sum_kernel_us_per_step": round(sum(v[1] for v in agg.values()) / n, 1), "launches_per_step": round(sum(v[0] for v in agg.values()) / n, 1)}
    print(json.dumps(out))
    for name, (c, t, grids) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %-70s %6.1f launches  %8.1f us/step  %7.1f us avg  grids %s" % (name, c / n, t / n, t / c, dict(grids.most_common(3))))


table([s[0] for s in steps], "backward")
table([s[1] for s in steps], "forward (of the next step)")
