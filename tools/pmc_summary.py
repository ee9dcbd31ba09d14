#!/usr/bin/env python3
"""Summary of rocprofv3 --pmc passes written by tools/gpu_trip.sh (step `pmc`): per counter the mean over the dispatches of the kernels
whose name contains a substring, the mean dispatch duration of pass 1, and the derived fractions the notebook quotes.
Usage: python tools/pmc_summary.py <out dir> <pass directory prefix> <kernel name substring>"""
import collections
import csv
import glob
import sys


def main():
    out, prefix, sub = sys.argv[1], sys.argv[2], sys.argv[3]
    agg = collections.defaultdict(list)
    dur = []
    grid = None
    for f in sorted(glob.glob(f"{out}/{prefix}*/**/*counter_collection.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                grid = r.get("Grid_Size", grid)
    for f in sorted(glob.glob(f"{out}/{prefix}1/**/*kernel_trace.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    d = sum(dur) / max(1, len(dur))
    print(f"kernel *{sub}*: {len(dur)} dispatches, mean duration under the profiler {d:.1f} us, grid {grid}")
    for k in sorted(m):
        print(f"  {k:34s} {m[k]:.5g}")
    # SQ_BUSY_CYCLES counts per XCD-SE...; the ratio the notebook uses: matrix-pipe busy cycles / (4 SIMDs x CUs x kernel cycles)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CYCLES" in m and m["SQ_BUSY_CYCLES"] > 0:
        print(f"  matrix pipe busy / (SQ_BUSY_CYCLES x CUs per SE share): see notebook; raw ratio MFMA_BUSY / BUSY = {m['SQ_VALU_MFMA_BUSY_CYCLES'] / m['SQ_BUSY_CYCLES']:.3f}")
    if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m and m["SQ_WAVE_CYCLES"] > 0:
        print(f"  waves parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES) = {m['SQ_WAIT_ANY'] / m['SQ_WAVE_CYCLES']:.3f}")
    if "TA_TA_BUSY_sum" in m and "TCP_GATE_EN1_sum" in m and m["TCP_GATE_EN1_sum"] > 0:
        print(f"  TA busy (TA_TA_BUSY_sum / TCP_GATE_EN1_sum) = {m['TA_TA_BUSY_sum'] / m['TCP_GATE_EN1_sum']:.3f}")
    if "TCP_PENDING_STALL_CYCLES_sum" in m and "TCP_GATE_EN1_sum" in m and m["TCP_GATE_EN1_sum"] > 0:
        print(f"  TCP pending stall fraction = {m['TCP_PENDING_STALL_CYCLES_sum'] / m['TCP_GATE_EN1_sum']:.3f}")


if __name__ == "__main__":
    main()
