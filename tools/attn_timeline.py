#!/usr/bin/env python3
"""Per-workgroup stamps of the experimental persistent fp32 attention kernel (libnomad_diag.so; run with NOMAD_ATTN_PIPE=1
NOMAD_ATTN_ABLATE=32, or 32 + other probe bits): wall time and shader
clock over a workgroup's life (-> the clock the CUs really run at under this kernel) and the SIMD each wave sits on."""
import collections, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
B, T = int(os.environ.get("B", 256)), int(os.environ.get("T", 199))
eng = Engine(seeded_state_dict(0), 0, diag=True)
qkv = (torch.randn(B * T, 2304, generator=torch.Generator().manual_seed(0)) * float(os.environ.get("SCALE", "0.5"))).cuda()
for _ in range(4):
    eng.diag_attention(qkv, B, T)
torch.cuda.synchronize()
n = 768
buf = (C.c_ulonglong * (6 * n))()
assert eng.lib.nomad_diag_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.int64)
wall_us = (t[:, 1] - t[:, 0]) / 100.0
clk = (t[:, 3] - t[:, 2])
simd = np.frombuffer(t[:, 5].astype(np.uint64).tobytes(), dtype=np.uint8).reshape(n, 8)[:, :4]
pat = collections.Counter(tuple(int(x) for x in row) for row in simd)
hw, xcc = t[:, 4] & 0xFFFFFFFF, (t[:, 4] >> 32) & 0xF
cu = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)
per_cu = collections.Counter(int(c) for c in cu)
print(json.dumps({"ablate": os.environ.get("NOMAD_ATTN_ABLATE"), "wg_life_us": {"mean": round(float(wall_us.mean()), 1), "min": round(float(wall_us.min()), 1), "max": round(float(wall_us.max()), 1)},
                  "start_spread_us": round(float((t[:, 0].max() - t[:, 0].min()) / 100.0), 1),
                  "kernel_span_us": round(float((t[:, 1].max() - t[:, 0].min()) / 100.0), 1),
                  "clock_per_wall_MHz": {"mean": round(float((clk / wall_us).mean()), 1), "min": round(float((clk / wall_us).min()), 1), "max": round(float((clk / wall_us).max()), 1)},
                  "wave_to_simd_patterns": {str(k): v for k, v in pat.most_common(6)},
                  "workgroups_per_cu": dict(collections.Counter(per_cu.values()))}))

life = wall_us
by_xcc = collections.defaultdict(list)
for i in range(n):
    by_xcc[int(xcc[i])].append(life[i])
print("mean life per XCC:", {k: round(float(np.mean(v)), 1) for k, v in sorted(by_xcc.items())})
groups = collections.defaultdict(list)
for i in range(n):
    groups[int(cu[i])].append(i)
rows = []
for c, idx in groups.items():
    rows.append((max(life[i] for i in idx), c, [(round(float(life[i]), 1), tuple(int(x) for x in simd[i]), int(i)) for i in idx]))
rows.sort()
for rrow in rows[:6] + rows[-6:]:
    print(hex(rrow[1]), rrow[2])
same = [len(set(tuple(int(x) for x in simd[i]) for i in idx)) for idx in groups.values()]
cu_max = np.array([max(life[i] for i in idx) for idx in groups.values()])
print("distinct wave->SIMD patterns among a CU's 3 workgroups -> mean of the CU's slowest life:", {k: round(float(cu_max[np.array(same) == k].mean()), 1) for k in sorted(set(same))}, collections.Counter(same))
