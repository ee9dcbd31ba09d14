#!/usr/bin/env python3
"""Where a 256x128 fp32 GEMM tile's time goes on its CU: per-workgroup wall-clock stamps of wave 0 (entry, first barrier passed,
K loop done, epilogue stores issued, stores acknowledged) from the timing-probe instantiations of gemm_f32_glds_kernel
(libnomad_diag.so, tile ids 68 = production, 69 = without the epilogue stores), grouped by CU (two workgroups resident per CU).
Usage: python tools/gemm_timeline_f32.py [--shapes qkv,fc1] [--tiles 68,69]"""
import argparse, collections, ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nomad_amd.engine import Engine
from nomad_amd.weights import seeded_state_dict
from gemm_sweep import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="qkv,fc1,fc2")
    ap.add_argument("--tiles", default="68,69")
    ap.add_argument("--dump", default=None, help="directory for the raw stamps (npy)")
    a = ap.parse_args()
    eng = Engine(seeded_state_dict(0), 0, diag=True)
    g = torch.Generator().manual_seed(0)
    for sname in a.shapes.split(","):
        M, N, K, has_b, gelu, has_r = SHAPES[sname]
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * K ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda() if has_b else None
        R = torch.randn(M, N, generator=g).cuda() if has_r else None
        for tile in (int(t) for t in a.tiles.split(",")):
            for _ in range(4):
                eng.diag_gemm(A, W, b, R, gelu=gelu, tile=tile)
            torch.cuda.synchronize()
            nwg = ((M + 255) // 256) * (N // 128)
            n = min(nwg, 4096)
            buf = (C.c_ulonglong * (6 * n))()
            assert eng.lib.nomad_diag_timeline(buf, n) == 0
            t = np.frombuffer(buf, dtype=np.uint64).reshape(n, 6).astype(np.int64)
            if a.dump:
                os.makedirs(a.dump, exist_ok=True)
                np.save(os.path.join(a.dump, f"timeline_{sname}_{tile}.npy"), t)
            us = (t[:, :5] - t[:, 0].min()) / 100.0                    # 100 MHz ticks -> us
            if tile in (86, 95):   # set-up detail build: {entry, tile coordinates known, offsets + bases done, accumulators zeroed (first DMA next), first barrier passed}
                later = us[:, 0] >= 1.0
                seg = {"entry_to_tile_coords_us": us[:, 1] - us[:, 0], "coords_to_offsets_done_us": us[:, 2] - us[:, 1],
                       "offsets_to_first_dma_us": us[:, 3] - us[:, 2], "first_dma_to_barrier_passed_us": us[:, 4] - us[:, 3]}
                print(json.dumps({"shape": sname, "tile": tile, "setup_detail": {k: {"first_round": round(float(v[~later].mean()), 2), "later": round(float(v[later].mean()), 2),
                                  "later_p90": round(float(np.percentile(v[later], 90)), 2)} for k, v in seg.items()}}), flush=True)
                continue
            if tile in (76, 81):   # prologue-detail build: {entry, address set-up done, own share of K tile 0 landed, first barrier passed, K loop done}
                later = us[:, 0] >= 1.0
                seg = {"entry_to_setup_done_us": us[:, 1] - us[:, 0], "first_dma_issue_to_landed_us": us[:, 2] - us[:, 1],
                       "landed_to_barrier_passed_us": us[:, 3] - us[:, 2], "k_loop_us": us[:, 4] - us[:, 3]}
                print(json.dumps({"shape": sname, "tile": tile, "prologue_detail": {k: {"first_round": round(float(v[~later].mean()), 2), "later": round(float(v[later].mean()), 2),
                                  "later_p90": round(float(np.percentile(v[later], 90)), 2)} for k, v in seg.items()}}), flush=True)
                continue
            pro, loop, epi, ack = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2], us[:, 4] - us[:, 3]
            hw, xcc = t[:, 5] & 0xFFFFFFFF, (t[:, 5] >> 32) & 0xF
            cu = (xcc << 16) | (((hw >> 13) & 7) << 12) | (((hw >> 12) & 1) << 8) | ((hw >> 8) & 0xF)   # (xcc, se, sh, cu)
            per_cu = collections.defaultdict(list)
            for i in range(n):
                per_cu[int(cu[i])].append(i)
            gaps, solo_frac, both_loop_frac, span = [], [], [], []
            loop_overlapped, loop_clean = [], []
            for ids in per_cu.values():
                ids.sort(key=lambda i: us[i, 0])
                ends = sorted(us[i, 4] for i in ids)
                # slot hand-over: every start after the first two takes the slot of the latest earlier end not yet taken
                free = []
                k = 0
                for i in ids[2:]:
                    while k < len(ends) and ends[k] <= us[i, 0] + 1e-9:
                        free.append(ends[k]); k += 1
                    if free:
                        gaps.append(us[i, 0] - free.pop(0))
                # occupancy of the CU's matrix pipes: time with 0 / 1 / 2 workgroups inside their K loop
                ev = sorted([(us[i, 1], 1) for i in ids] + [(us[i, 2], -1) for i in ids])
                t0c, t1c = us[ids[0], 0], max(us[i, 4] for i in ids)
                cur, last, acc = 0, t0c, [0.0, 0.0, 0.0]
                for tt, d in ev:
                    acc[min(cur, 2)] += tt - last
                    last, cur = tt, cur + d
                acc[min(cur, 2)] += t1c - last
                tot = t1c - t0c
                solo_frac.append((acc[0] / tot, acc[1] / tot, acc[2] / tot))
                span.append(tot)
                # K loops that ran while another workgroup of the CU was in its epilogue / hand-over vs loops that did not
                for i in ids:
                    other_epi = sum(max(0.0, min(us[i, 2], us[j, 4]) - max(us[i, 1], us[j, 2])) for j in ids if j != i)
                    (loop_overlapped if other_epi > 0.5 else loop_clean).append((loop[i], other_epi))
            sf = np.array(solo_frac)
            first = us[:, 0] < 1.0
            res = {"shape": sname, "tile": tile, "M": M, "N": N, "K": K, "workgroups": nwg, "stamped": n, "distinct_cus": len(per_cu),
                   "stamped_span_us": round(float(us[:, 4].max()), 1),
                   "entry_to_first_barrier_us": {"first_round": round(float(pro[first].mean()), 2), "later": round(float(pro[~first].mean()), 2) if (~first).any() else None},
                   "k_loop_us": {"mean": round(float(loop.mean()), 2), "p10": round(float(np.percentile(loop, 10)), 2), "p90": round(float(np.percentile(loop, 90)), 2)},
                   "epilogue_issue_us": round(float(epi.mean()), 2), "store_ack_wait_us": round(float(ack.mean()), 2),
                   "slot_handover_gap_us": {"mean": round(float(np.mean(gaps)), 2), "median": round(float(np.median(gaps)), 2), "p90": round(float(np.percentile(gaps, 90)), 2)} if gaps else None,
                   "cu_time_fraction_with_0_1_2_workgroups_in_k_loop": [round(float(x), 4) for x in sf.mean(axis=0)],
                   "k_loop_us_when_peer_in_epilogue": round(float(np.mean([x[0] for x in loop_overlapped])), 2) if loop_overlapped else None,
                   "peer_epilogue_overlap_us": round(float(np.mean([x[1] for x in loop_overlapped])), 2) if loop_overlapped else None,
                   "k_loop_us_otherwise": round(float(np.mean([x[0] for x in loop_clean])), 2) if loop_clean else None}
            print(json.dumps(res), flush=True)
        del A, W, b, R
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
