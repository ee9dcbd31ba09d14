// libnomad_hip.so, translation unit 3 of 3: every bf16 / bf16x3 GEMM instantiation (gemm_bf16*.hip.h) and the code that picks one.
// (Split out of nomad_hip.hip in round 6: see nomad_ctx.hip.h.)
#include "nomad_ctx.hip.h"

#include "gemm_bf16.hip.h"
#include "gemm_bf16_8phase.hip.h"
#include "gemm_bf16_p9.hip.h"
#include "gemm_bf16x3.hip.h"

// what the PLAIN instantiations (small epilogue, lean set-up) require
static bool p8_plain_cr(const GemmParams& p) {
    return p.cmap.clip_rows >= p.M && !p.cmap.pref && p.c_colblk == 0 && p.c_blk_step == 0 &&
           (!p.R || (p.rmap.clip_rows >= p.M && !p.rmap.pref)) &&
           !p.amap.pref && p.group_m == 0 && (p.amap.clip_rows >= p.M || p.amap.clip_rows >= 2);
}

// what the persistent 256 x 256 kernel (gemm_bf16_p9.hip.h) requires on top of p8_plain_cr: one group, contiguous K, every column
// stored, no split planes, not GELU and residual together
static bool p9_applies(const GemmParams& p, int groups) {
    return groups == 1 && p8_plain_cr(p) && p.N % 256 == 0 && p.n_valid == p.N && p.K % 128 == 0 && p.kchunk == p.K &&
           p.a_plane == 0 && p.c_plane == 0 && !p.Upre && !p.DG &&
           !(p.gelu && p.R);   // (GELU and a residual in one epilogue: no GEMM of the model has both, and the shipped instantiation has no copy for it)
}

// 256 x 192 instead of 256 x 256 tiles in the deep-pipelined bf16 kernel (gemm_bf16_8phase.hip.h, NJ = 3) for the N = 768 GEMMs
// of config C5 (out_proj, fc2: 188 row tiles x 3 = 2.2 rounds of the 256 CUs, 2.94 with 192-column tiles).  Bit-identical
// results.  OFF by default: in isolation fc2 runs 19 % and out_proj 12 % faster (hipBLASLt picks MT256x192 there too), but inside
// the C5 forward the other GEMMs slow down by more than that - the chip holds 2130 instead of 2157 MHz (2400 nominal) with
// them, 19.25 vs 18.98 ms per forward (profiles/r03_n192_null.txt).  NOMAD_BF16_N192=1 takes them wherever they save a round
// (a 192-column tile costs ~0.78 of a 256-column one), =2 wherever N % 192 == 0.
static bool p8_use_n192(const nomad_ctx* c, int M, int N, int K) {
    const int mode = c->tune.p8_n192;
    if (N % 192 != 0 || mode <= 0) return false;
    if (mode == 2) return true;
    if (mode == 3 && K < 2048) return false;   // A/B: the long-K problems only (fc2)
    if (mode == 4 && K >= 2048) return false;  // A/B: the short-K problems only (out_proj, proj)
    const long long tm = (M + 255) / 256;
    const long long r256 = (tm * (N / 256) + 255) / 256, r192 = (tm * (N / 192) + 255) / 256;
    return 0.80 * (double)r192 < 0.95 * (double)r256;
}

int run_gemm_bf16(nomad_ctx* c, GemmParams p, int groups, hipStream_t s, int tile) {
    // Tuning (A/B switches of the diag library; defaults = shipped): p8_min_tiles - smallest grid in 256 x 256 tiles that takes the
    // deep-pipelined kernels; p8_nt_stores - their output stores carry the non-temporal hint; p8_rpre - residual prefetch in p8_epilogue
    // (0 off, 1 residual GEMMs only, 2 every GEMM, 3 = 2 + the small epilogue for plain C / R); x3_plain_epi - one bf16x3 instantiation
    // with a run-time output format; p8_three_b - three B buffers (160 KB of LDS); p9 - the persistent kernel wherever it applies
    const Tuning& tu = c->tune;
    auto p8_min_tiles = [&] { return tu.p8_min_tiles; };
    auto p8_nt_stores = [&] { return tu.p8_nt_stores; };
    auto p8_residual_prefetch = [&] { return tu.p8_rpre; };
    auto x3_plain_epilogue = [&] { return tu.x3_plain_epi; };
    auto p8_three_b = [&] { return tu.p8_three_b; };
    auto p9_on = [&] { return tu.p9; };
    const double flops = 2.0 * p.M * (double)p.n_valid * p.K * groups;  // bf16x3: the fp32-equivalent count, not 3x
    if (tile < 0) {
        // measured (profiles/r01_gemm_sweep_bf16.json): 256x256 tiles (wave tile 64x128) win on wide (N >= 1024)
        // and very tall problems, 128x128 (8 waves) on the N = 768 / 512 transformer shapes
        if (p.N % 128 != 0) tile = p.M < 512 ? 4 : 2;
        else if (p.M < 512) tile = 4;
        else if (p.N % 256 == 0 && p.K % 128 == 0 && groups == 1 && (long long)((p.M + 255) / 256) * (p.N / 256) >= p8_min_tiles())
            tile = (p9_on() && p9_applies(p, groups) && (tu.p9_res || !p.R)) ? 60 : (p8_three_b() && p8_nt_stores() && p8_use_n192(c, p.M, p.N, p.K)) ? 55 : 16;  // deep-pipelined 256x256 / 256x192 kernel once there are >= 2 rounds of tiles (profiles/r01_gemm_sweep_bf16_8phase.json)
        else tile = (p.N % 256 == 0 && (p.N >= 1024 || p.M >= 100000)) ? 3 : 1;
    }
    Scope sc(c, s, NOMAD_K_GEMM, flops, (tile == 1 || tile == 3 || tile == 16 || tile == 60 || tile == 61 || tile == 62 || tile == 63 || tile == 64 || tile == 68 || tile == 65 || tile == 66 || tile == 67 || tile == 55 || tile == 56 || tile == 57 || tile == 58 || tile == 42 || tile == 43 || tile == 44 || tile == 45 || tile == 46 || tile == 47 || tile == 48 || tile == 49 || tile == 50 || tile == 51 || tile == 52 || tile == 53 || tile == 54 || tile == 20 || tile == 21 || tile == 27 || tile == 28 || tile == 32 || tile == 33) ? NOMAD_K_GEMM_BIG : (tile == 2 ? NOMAD_K_GEMM_FINE : -1));
    hipError_t e;
    switch (tile) {
        // the instantiations the bf16 / bf16x3 forwards select
        case 1: e = launch_gemm_bf16<128, 128, 4, 2>(p, groups, s); break;
        case 2: e = launch_gemm_bf16<128, 64, 4, 2>(p, groups, s); break;
        case 3: e = launch_gemm_bf16<256, 256, 4, 2>(p, groups, s); break;
        case 4: e = launch_gemm_bf16<64, 64, 2, 2>(p, groups, s); break;
        case 16:  // 256x256 deep-pipelined schedule (gemm_bf16_8phase.hip.h)
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = !p8_nt_stores() ? launch_gemm_bf16_8phase<0>(p, groups, s)
                : !p8_three_b() ? launch_gemm_bf16_8phase<8>(p, groups, s)
                : (p8_residual_prefetch() == 3 && p8_plain_cr(p)) ? launch_gemm_bf16_8phase<8, false, 0, 3, 4, true, true>(p, groups, s)
                : ((p.R && p8_residual_prefetch() == 1) || p8_residual_prefetch() >= 2) ? launch_gemm_bf16_8phase<8, false, 0, 3, 4, true>(p, groups, s)
                                                  : launch_gemm_bf16_8phase<8, false, 0, 3>(p, groups, s);
            break;
        case 60:  // persistent form of the deep-pipelined kernel: one workgroup per CU walks tiles, direct epilogue (gemm_bf16_p9.hip.h)
        case 64: {  // ... (64: never split by rows - A/B)
            if (!p9_applies(p, groups)) return fail(NOMAD_ERR_INVALID, "bf16 persistent gemm: plain C / R, one group, N %% 256, K %% 128, contiguous K");
            // Tile quantisation (round 5).  One workgroup per CU and tiles of 256 x 256: the N = 768 GEMMs of config C5 (out_proj, fc2) are
            // 564 tiles = 2.2 rounds of the 256 CUs, i.e. three rounds' time.  The one-tile-per-workgroup kernel hid that behind the
            // OTHER half of the batch on a second stream; persistent workgroups of two launches cannot share CUs.  Instead the rows of
            // the whole rounds go to the persistent kernel and the rows of the sparse last round to the 128 x 128 kernel (tile 1,
            // two workgroups per CU) right behind it on the same stream: every bf16 kernel contracts k in the same order, so which
            // kernel computes a row changes no bit (tests/test_gpu_bf16.py).  Plain A matrices only (the conv stack's per-clip maps
            // have thousands of tiles); Tuning::p9_tail_split = 0 switches it off.
            // (Tuning::p9_share, A/B: with n concurrent parts of a batch on n streams each launch takes 1 / n of the CUs, so that the parts'
            // persistent launches run side by side instead of queueing for each other's LDS)
            const int cus = (tu.p9_share && tu.concurrent_parts > 1) ? std::max(8, c->num_cus / tu.concurrent_parts) : c->num_cus;
            const int grid = 8 * std::max(1, cus / 8);
            const long long tn = p.N / 256, tm = (p.M + 255) / 256, tiles = tm * tn;
            const long long rounds = tiles / grid, rem = tiles - rounds * grid;
            if (tile == 60 && tu.p9_tail_split && rounds >= 1 && rounds <= 4 && rem > 0 && rem * 10 < grid * 6 && p.amap.clip_rows >= p.M && p.N % 128 == 0) {
                const int m_main = (int)(rounds * grid / tn) * 256;
                if (m_main > 0 && m_main < p.M) {
                    GemmParams a = p, b = p;
                    a.M = m_main;
                    a.amap = plain_map(a.M, p.amap.ld); a.amap.off = p.amap.off;
                    a.cmap = plain_map(a.M, p.cmap.ld); a.cmap.off = p.cmap.off;
                    a.rmap = plain_map(a.M, p.rmap.ld); a.rmap.off = p.rmap.off;
                    b.M = p.M - m_main;
                    b.amap = plain_map(b.M, p.amap.ld); b.amap.off = p.amap.off + (long long)m_main * p.amap.ld;
                    b.cmap = plain_map(b.M, p.cmap.ld); b.cmap.off = p.cmap.off + (long long)m_main * p.cmap.ld;
                    b.rmap = plain_map(b.M, p.rmap.ld); b.rmap.off = p.rmap.off + (long long)m_main * p.rmap.ld;
                    e = launch_gemm_bf16_p9<0, true>(a, s, cus);
                    if (e == hipSuccess) e = launch_gemm_bf16<128, 128, 4, 2>(b, groups, s);
                    break;
                }
            }
            // Short (192-row) tiles, round 6: a run-time mode of the same instantiation.  A 192 x 256 tile costs ~0.80 of a 256 x 256 one (three
            // quarters of the MFMAs, 7 / 8 of the LDS-DMA bytes); it is taken where the largest tile count any CU gets, priced so, is smaller.
            // Only for a batch that runs ALONE (concurrent_parts == 1): next to the other half of a two-stream batch the CUs never idle - the
            // other half's workgroups take a CU the moment a workgroup leaves it - so what counts there is the total work, which short tiles
            // raise (measured, gpurun_out/r6a: two streams 1888 -> 1877 clips/s with short tiles, one stream 1828-1845 -> 1876-1878).
            if (tile == 60 && (tu.p9_short == 2 || (tu.p9_short == 1 && tu.concurrent_parts <= 1))) {
                const long long tm_s = (p.M + 191) / 192;
                const long long r_full = (tiles + grid - 1) / grid, r_short = (tm_s * tn + grid - 1) / grid;
                p.p9_short = (tu.p9_short == 2 || 0.80 * (double)r_short < 0.95 * (double)r_full) ? 1 : 0;
            }
            p.p9_late = tu.p9_late ? 1 : 0;
            p.p9_wl = tu.p9_wl ? 1 : 0;
            e = launch_gemm_bf16_p9<0, true>(p, s, cus);
            break;
        }
        case 57:  // the deep-pipelined kernel with the residual prefetch in the epilogue, general C / R addressing
        case 58:  // ... with the small epilogue for plain C / R matrices (what tile 16 resolves to for them)
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            if (tile == 58 && !p8_plain_cr(p)) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm, plain epilogue: C / R are not plain matrices");
            e = tile == 57 ? launch_gemm_bf16_8phase<8, false, 0, 3, 4, true>(p, groups, s) : launch_gemm_bf16_8phase<8, false, 0, 3, 4, true, true>(p, groups, s);
            break;
        case 55:  // 256x192 tiles of the same schedule (three B buffers, nt stores)
            if (p.N % 192 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm, 192-column tiles: N %% 192, K %% 128");
            e = (p8_residual_prefetch() == 3 && p8_plain_cr(p)) ? launch_gemm_bf16_8phase<8, false, 0, 3, 3, true, true>(p, groups, s)
                                                                : launch_gemm_bf16_8phase<8, false, 0, 3, 3>(p, groups, s);
            break;
        case 27:  // bf16x3, every plane staged once (gemm_bf16x3.hip.h): split output
        case 28:  // ... fp32 output
            if (p.N % 256 != 0 || p.K % 64 != 0) return fail(NOMAD_ERR_INVALID, "bf16x3 gemm: N %% 256, K %% 64");
            if (p8_nt_stores() && x3_plain_epilogue() && p8_plain_cr(p) && (tile == 27) == (p.c_plane != 0))
                e = launch_gemm_bf16x3<8, 3, 2, true>(p, groups, s);   // one instantiation for both output formats, small epilogue
            else if (p8_nt_stores()) e = tile == 27 ? launch_gemm_bf16x3<8, 1>(p, groups, s) : launch_gemm_bf16x3<8, 2>(p, groups, s);
            else e = tile == 27 ? launch_gemm_bf16x3<0, 1>(p, groups, s) : launch_gemm_bf16x3<0, 2>(p, groups, s);
            break;
#ifdef NOMAD_DIAG
        // experimental instantiations, cross-check kernels and timing probes (libnomad_diag.so)
        case 0: e = launch_gemm_bf16<256, 128, 4, 2>(p, groups, s); break;
        case 5: e = launch_gemm_bf16<128, 128, 2, 2>(p, groups, s); break;
        case 6: e = launch_gemm_bf16<256, 128, 2, 2>(p, groups, s); break;
        case 7: e = launch_gemm_bf16<128, 128, 4, 2, 1>(p, groups, s); break;   // ablation: no epilogue stores
        case 8: e = launch_gemm_bf16<128, 128, 4, 2, 2>(p, groups, s); break;   // ablation: one K tile only
        case 9: e = launch_gemm_bf16<256, 256, 2, 4, 0, 32, 4>(p, groups, s); break;   // wave 128x64, BK 32, 4-stage
        case 10: e = launch_gemm_bf16<256, 256, 4, 2, 0, 32, 4>(p, groups, s); break;  // wave 64x128
        case 11: e = launch_gemm_bf16<128, 128, 4, 2, 0, 32, 4>(p, groups, s); break;
        case 12: e = launch_gemm_bf16<128, 128, 4, 2, 0, 64, 3>(p, groups, s); break;
        case 13: e = launch_gemm_bf16<256, 128, 4, 2, 0, 64, 3>(p, groups, s); break;
        case 14: e = launch_gemm_bf16<256, 128, 4, 2, 0, 32, 4>(p, groups, s); break;
        case 15: e = launch_gemm_bf16<256, 256, 2, 4, 0, 64, 2>(p, groups, s); break;
        case 17:  // 8-phase ablation: no epilogue stores
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = launch_gemm_bf16_8phase<1>(p, groups, s);
            break;
        case 18:  // A/B: 8-phase kernel with buffer_load..lds
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = launch_gemm_bf16_8phase<0, true>(p, groups, s);
            break;
        case 19:  // A/B: 8-phase kernel without s_setprio
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = launch_gemm_bf16_8phase<2>(p, groups, s);
            break;
        case 20:  // bf16x3 cross-check (K-concatenated operands in the 8-phase kernel): split output
        case 21:  // ... fp32 output
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16x3 gemm: N %% 256, K %% 128");
            e = tile == 20 ? launch_gemm_bf16_8phase<0, false, 1>(p, groups, s) : launch_gemm_bf16_8phase<0, false, 2>(p, groups, s);
            break;
        case 42:  // A/B: 8-phase kernel with non-temporal output stores / + residual loads / residual loads only
        case 43:
        case 44:
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = tile == 42 ? launch_gemm_bf16_8phase<8>(p, groups, s) : tile == 43 ? launch_gemm_bf16_8phase<9>(p, groups, s) : launch_gemm_bf16_8phase<10>(p, groups, s);
            break;
        case 56:  // 256x192 tiles with two B buffers (A/B against 55)
            if (p.N % 192 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm, 192-column tiles: N %% 192, K %% 128");
            e = launch_gemm_bf16_8phase<8, false, 0, 2, 3>(p, groups, s);
            break;
        case 46:  // three B buffers: B staged 1.75 K tiles ahead (non-temporal stores as tile 16 ships them)
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = launch_gemm_bf16_8phase<8, false, 0, 3>(p, groups, s);
            break;
        case 47:  // timing probes on the plain bf16 kernel (wrong results): no LDS-DMA / neither DMA nor LDS reads / no LDS reads
        case 48:
        case 49:
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = tile == 47 ? launch_gemm_bf16_8phase<4>(p, groups, s) : tile == 48 ? launch_gemm_bf16_8phase<5>(p, groups, s) : launch_gemm_bf16_8phase<6>(p, groups, s);
            break;
        case 51:  // cache-policy probes of the LDS-DMA on the shipped kernel (three B buffers, nt stores): nt on A / B / both, sc1 on both
        case 52:
        case 53:
        case 54:
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = tile == 51 ? launch_gemm_bf16_8phase<13, false, 0, 3>(p, groups, s) : tile == 52 ? launch_gemm_bf16_8phase<14, false, 0, 3>(p, groups, s)
              : tile == 53 ? launch_gemm_bf16_8phase<15, false, 0, 3>(p, groups, s) : launch_gemm_bf16_8phase<16, false, 0, 3>(p, groups, s);
            break;
        case 50:  // timing probe: no loads and no barriers in the loop (both wave rows issue MFMAs at once)
            e = launch_gemm_bf16_8phase<12>(p, groups, s);
            break;
        case 45:  // timing probe: every workgroup stages A tile 0 (wrong results)
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = launch_gemm_bf16_8phase<11>(p, groups, s);
            break;
        case 36:  // timing probe: per-workgroup timeline (nomad_diag_timeline, tools/gemm_timeline.py)
            if (p.N % 256 != 0 || p.K % 128 != 0) return fail(NOMAD_ERR_INVALID, "bf16 8-phase gemm: N %% 256, K %% 128");
            e = launch_gemm_bf16_8phase<7>(p, groups, s);
            break;
        case 65:  // persistent kernel, B DMA of a K tile issued in phase 3 / in phases 2 and 3 / every DMA inside an MFMA cluster (A/B against 60)
        case 66:
        case 67:
            if (!p9_applies(p, groups)) return fail(NOMAD_ERR_INVALID, "bf16 persistent gemm: plain C / R, one group, N %% 256, K %% 128, contiguous K");
            e = tile == 65 ? launch_gemm_bf16_p9<0, true, 1>(p, s, c->num_cus) : tile == 66 ? launch_gemm_bf16_p9<0, true, 2>(p, s, c->num_cus)
                                                                                      : launch_gemm_bf16_p9<0, true, 3>(p, s, c->num_cus);
            break;
        case 68:  // persistent kernel, 192-row tiles forced (A/B against 64 = never short; 60 = the shipped choice)
            if (!p9_applies(p, groups)) return fail(NOMAD_ERR_INVALID, "bf16 persistent gemm: plain C / R, one group, N %% 256, K %% 128, contiguous K");
            p.p9_short = 1;
            p.p9_late = tu.p9_late ? 1 : 0;
            p.p9_wl = tu.p9_wl ? 1 : 0;
            e = launch_gemm_bf16_p9<0, true>(p, s, c->num_cus);
            break;
        case 61:  // persistent kernel: per-workgroup timeline probe / no output stores (timing) / every epilogue between tiles (A/B)
        case 62:
        case 63:
            if (!p9_applies(p, groups)) return fail(NOMAD_ERR_INVALID, "bf16 persistent gemm: plain C / R, one group, N %% 256, K %% 128, contiguous K");
            if (tile == 61) p.p9_skew = tu.p9_skew;
            if (tile != 63) { p.p9_late = tu.p9_late ? 1 : 0; p.p9_wl = tu.p9_wl ? 1 : 0; }   // (63 keeps the accumulator-shaped stores: the bit-identity tests cover both)
            e = tile == 61 ? launch_gemm_bf16_p9<7, true>(p, s, c->num_cus) : tile == 62 ? launch_gemm_bf16_p9<1, true>(p, s, c->num_cus)
                                                                                      : launch_gemm_bf16_p9<0, false>(p, s, c->num_cus);
            break;
        case 22: e = launch_gemm_bf16_8phase<3, false, 2>(p, groups, s); break;  // bf16x3 timing probes, fp32 output
        case 23: e = launch_gemm_bf16_8phase<4, false, 2>(p, groups, s); break;
        case 24: e = launch_gemm_bf16_8phase<5, false, 2>(p, groups, s); break;
        case 25: e = launch_gemm_bf16_8phase<6, false, 2>(p, groups, s); break;
        case 26: e = launch_gemm_bf16_8phase<1, false, 2>(p, groups, s); break;
        case 29:  // timing probe: no epilogue stores
        case 30:  // timing probe: no LDS-DMA
        case 31:  // timing probe: every workgroup stages A tile 0 (A always hits in L2)
            if (p.N % 256 != 0 || p.K % 64 != 0) return fail(NOMAD_ERR_INVALID, "bf16x3 gemm: N %% 256, K %% 64");
            e = tile == 29 ? launch_gemm_bf16x3<1, 2>(p, groups, s) : tile == 30 ? launch_gemm_bf16x3<4, 2>(p, groups, s)
                                                                                  : launch_gemm_bf16x3<7, 2>(p, groups, s);
            break;
        case 32:  // three A buffers (K % 192 == 0): split output
        case 33:  // ... fp32 output
            if (p.N % 256 != 0 || p.K % 192 != 0) return fail(NOMAD_ERR_INVALID, "bf16x3 gemm (3 A buffers): N %% 256, K %% 192");
            e = tile == 32 ? launch_gemm_bf16x3<0, 1, 3>(p, groups, s) : launch_gemm_bf16x3<0, 2, 3>(p, groups, s);
            break;
#endif
        default: return fail(NOMAD_ERR_INVALID, "bf16 gemm tile id %d is not in this library (experimental instantiations live in libnomad_diag.so)", tile);
    }
    if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "bf16 gemm launch: %s", hipGetErrorString(e));
    return 0;
}

#ifdef NOMAD_DIAG
int gemm_bf16_timeline_read(unsigned long long* out_host, int n) {
    HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 6 * (size_t)n));
    return 0;
}
#endif
