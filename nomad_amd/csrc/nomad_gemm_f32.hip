// libnomad_hip.so, translation unit 2 of 3: every fp32 GEMM instantiation (gemm_f32.hip.h) and the code that picks one.
// (Split out of nomad_hip.hip in round 6 so that the three units compile side by side: see nomad_ctx.hip.h.)
#include "nomad_ctx.hip.h"

int gemm_f32_dispatch(nomad_ctx* c, GemmParams p, int groups, int tile, hipStream_t s, int occ) {
    if (tile == 29 && occ == 0) occ = 4;  // measured: 4 workgroups/CU is the best residency for the 128x64x32 kernel
    const double flops = 2.0 * p.M * (double)p.n_valid * p.K * groups;
    Scope sc(c, s, NOMAD_K_GEMM, flops, tile == 33 ? NOMAD_K_GEMM_BIG : ((tile == 34 || tile == 31 || tile == 48) ? NOMAD_K_GEMM_FINE : -1));
    hipError_t e;
    if (c->gemm_x3 && tile == 48 && p.N == 64)
        tile = 37;  // bf16x3 products: the grouped pos-conv on the generic 64 x 64 tile (N padded 48 -> 64: a quarter of the MFMAs
                    // wasted, but 6 bf16 MFMAs of 32 cycles per 32-deep k range instead of 48 fp32 ones of 32 in the N = 48 kernel)
    if (c->gemm_x3 && (tile == 31 || tile == 34 || tile == 37) && p.N % 128 == 0 &&
        (long long)((p.M + 127) / 128) * (p.N / 128) >= 512)
        tile = 20;  // bf16x3 products: a 64 x 64 wave tile (128 x 128, 4 waves) does 12 MFMAs per 4 fragment splits where the
                    // 32 x 32 one does 3 per 2 - the split is VALU work - so it is taken as soon as it fills two rounds of CUs
    // plain C / R (/ Upre / DG) matrices: the instantiations with the small, residual-prefetching epilogue (gemm_f32.hip.h, OPT bits
    // 16 / 32) - every GEMM of the uniform scoring forward but the pos-conv's neighbours.  NOMAD_F32_PLAIN_EPI=0: the general
    // epilogue (A/B runs)
    const bool plain_cr = c->tune.f32_plain_epi && p.c_colblk == 0 && p.cmap.clip_rows >= p.M && !p.cmap.pref &&
                          (!p.R || (p.rmap.clip_rows >= p.M && !p.rmap.pref)) && (!p.DG || (p.dgmap.clip_rows >= p.M && !p.dgmap.pref));
    if (c->gemm_x3 && plain_cr && (tile == 20 || tile == 31 || tile == 34 || tile == 37)) {   // (not 33: its X3 form needs 146 VGPRs)
        constexpr int T = 16 | 32;   // one plain instantiation per tile for scoring and training alike (this mode is the small-batch / training one)
        switch (tile) {
            case 20: e = launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12 | T, true>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES)); break;
            case 31: e = launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | T, true>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)); break;
            case 34: e = launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12 | T, true>(p, groups, s); break;
            default: e = launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12 | T, true>(p, groups, s); break;
        }
        if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm (bf16x3 products) launch: %s", hipGetErrorString(e));
        return 0;
    }
    if (c->gemm_x3 && (tile == 20 || tile == 31 || tile == 33 || tile == 34 || tile == 37)) {
        // bf16x3 products on the same fp32 operands (nomad_set_gemm_precision): same tiles, staging and epilogues
        switch (tile) {
            case 20: e = launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12, true>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES)); break;
            case 31: e = launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12, true>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)); break;
            case 33: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13, true>(p, groups, s); break;
            case 34: e = launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12, true>(p, groups, s); break;
            default: e = launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12, true>(p, groups, s); break;
        }
        if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm (bf16x3 products) launch: %s", hipGetErrorString(e));
        return 0;
    }
    if (plain_cr && (tile == 33 || tile == 31 || tile == 37 || tile == 34 || tile == 20)) {
        constexpr int P = 16, T = 16 | 32;   // plain epilogue; + the training side operands (Upre / DG)
        const bool tr = p.Upre || p.DG;
        // Round 4 (gemm_f32.hip.h OPT bits 16384 / 64 / 1024; profiles/r04_gemm_f32_variants.txt): the scoring GEMMs on uniform
        // clip maps take the lean set-up (magic-number divisions on the scalar unit), the 256 x 128 tile also the skewed K
        // loop, and GEMMs without a residual the direct epilogue from transposed accumulators.  All bit-identical to the plain
        // instantiations.  NOMAD_F32_LEAN=0 / NOMAD_F32_DIRECT_EPI=0 / NOMAD_F32_SKEW=0 switch them off (A/B runs).
        const int variants = (c->tune.f32_lean ? 1 : 0) | (c->tune.f32_direct_epi ? 2 : 0) | (c->tune.f32_skew ? 4 : 0) | (c->tune.f32_res_ahead ? 8 : 0);
        // (a divisor of 1 - clips of ONE row, the shortest legal input - has no 32-bit magic number: those stay on the general set-up)
        const bool lean = (variants & 1) && !tr && !p.amap.pref && p.group_m == 0 && p.kchunk >= p.K &&
                          (p.amap.clip_rows >= p.M || p.amap.clip_rows >= 2);
        const bool direct = lean && (variants & 2) && !p.R && p.n_valid == p.N;
        const bool skew = lean && (variants & 4);
        // residual GEMMs (out_proj, fc2; no GELU): the residual of the next slab loaded ahead of the current slab's stores (OPT bit 32768)
        const bool ahead = lean && (variants & 8) && p.R && !p.gelu && p.n_valid == p.N && p.rmap.clip_rows >= p.M;
        constexpr int L = 16384, D = 1024, S = 64, RA = 32768;
        // two tile shapes in one launch when the last round of 256 x 128 tiles would be sparsely filled
        if (lean && skew && tile == 33 && groups == 1 && (p.R ? ahead : direct)) {
            // (sending the launches that need no split through the same kernel as well - one instantiation less alternating between
            // the launches of a transformer layer - changes nothing: 2405 vs 2408 clips/s)
            const int m1 = mixed_split_rows(c, p.M, p.N);
            if (m1 > 0) {
                e = p.R ? launch_gemm_mixed<13 | P | L | S | RA, 13 | P | L | S | RA>(p, m1, s)
                        : launch_gemm_mixed<13 | P | L | S | D, 13 | P | L | S | D>(p, m1, s);
                if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
                return 0;
            }
        }
        if (ahead && tile == 33 && skew) {
            e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | P | L | S | RA>(p, groups, s);
            if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
            return 0;
        }
        if (ahead && tile == 31) {
            e = launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | P | L | RA>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES));
            if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
            return 0;
        }
        if (lean && tile == 33) {
            e = direct ? (skew ? launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | P | L | S | D>(p, groups, s) : launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | P | L | D>(p, groups, s))
                       : (skew ? launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | P | L | S>(p, groups, s) : launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | P | L>(p, groups, s));
            if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
            return 0;
        }
        if (lean && tile == 31) {
            e = direct ? launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | P | L | D>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES))
                       : launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | P | L>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES));
            if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
            return 0;
        }
        if (lean && (tile == 37 || tile == 34 || tile == 20)) {   // the small-problem tiles (batch 1 .. config C4): one round of workgroups, the set-up is a visible part of each
            switch (tile) {
                case 37: e = direct ? launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12 | P | L | D>(p, groups, s) : launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12 | P | L>(p, groups, s); break;
                case 34: e = direct ? launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12 | P | L | D>(p, groups, s) : launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12 | P | L>(p, groups, s); break;
                default: e = direct ? launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12 | P | L | D>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES))
                                    : launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12 | P | L>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES)); break;
            }
            if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
            return 0;
        }
        switch (tile) {
            case 33: e = tr ? launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | T>(p, groups, s) : launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | P>(p, groups, s); break;
            case 31: e = tr ? launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | T>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES))
                            : launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | P>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)); break;
            case 20: e = tr ? launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12 | T>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES))
                            : launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12 | P>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES)); break;
            case 34: e = tr ? launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12 | T>(p, groups, s) : launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12 | P>(p, groups, s); break;
            default: e = tr ? launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12 | T>(p, groups, s) : launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12 | P>(p, groups, s); break;
        }
        if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
        return 0;
    }
    switch (tile) {
        // the instantiations pick_tile() / the pos-conv can select
        case 20: e = launch_gemm_glds<128, 128, 32, 2, 2, 2, false, 12>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 2, 2>::LDS_BYTES)); break;
        case 31: e = launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)); break;
        case 33: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13>(p, groups, s); break;   // 3-stage LDS-DMA pipeline (buffer_load..lds, issued mid-cluster), counted vmcnt
        case 34: e = launch_gemm_glds<128, 64, 32, 4, 2, 3, false, 12>(p, groups, s); break;
        case 37: e = launch_gemm_glds<64, 64, 32, 2, 2, 3, false, 12>(p, groups, s); break;
        case 48: e = launch_gemm_n48<true>(p, groups, s); break;    // N = 48 exactly (16x16x4 MFMA): the grouped pos-conv
#ifdef NOMAD_DIAG
        // experimental instantiations and ablations (libnomad_diag.so; tools/gemm_sweep.py, tests of the experimental tiles)
        case 0: e = launch_gemm<128, 128, 32, 2, 2>(p, groups, s); break;
        case 1: e = launch_gemm<128, 64, 16, 2, 2>(p, groups, s); break;
        case 2: e = launch_gemm<64, 64, 32, 2, 2>(p, groups, s); break;
        // experimental instantiations (tools/gemm_sweep.py)
        case 3: e = launch_gemm<128, 128, 16, 2, 2>(p, groups, s); break;
        case 4: e = launch_gemm<256, 128, 32, 4, 2>(p, groups, s); break;
        case 5: e = launch_gemm<256, 256, 32, 4, 2>(p, groups, s); break;
        case 6: e = launch_gemm<256, 128, 16, 4, 2>(p, groups, s); break;
        case 7: e = launch_gemm<128, 256, 32, 2, 2>(p, groups, s); break;
        case 8: e = launch_gemm<256, 256, 16, 4, 2>(p, groups, s); break;
        case 9: e = launch_gemm<128, 128, 16, 4, 2>(p, groups, s, 16 * 1024); break;   // 8 waves, forced 2 WG/CU
        case 10: e = launch_gemm<128, 128, 16, 2, 4>(p, groups, s, 16 * 1024); break;
        case 11: e = launch_gemm<128, 128, 32, 4, 2>(p, groups, s); break;
        case 12: e = launch_gemm<128, 128, 32, 2, 4>(p, groups, s); break;
        case 13: e = launch_gemm<128, 128, 16, 4, 2>(p, groups, s); break;               // 3 WG/CU if registers allow
        case 21: e = launch_gemm_glds<256, 128, 16, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 128, 16, 4, 2>::LDS_BYTES)); break;
        case 22: e = launch_gemm_glds<128, 128, 16, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 16, 4, 2>::LDS_BYTES)); break;
        case 23: e = launch_gemm_glds<256, 128, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 128, 32, 4, 2>::LDS_BYTES)); break;
        case 24: e = launch_gemm_glds<256, 256, 16, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 256, 16, 4, 2>::LDS_BYTES)); break;
        case 25: e = launch_gemm_glds<256, 256, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<256, 256, 32, 4, 2>::LDS_BYTES)); break;
        case 26: e = launch_gemm_glds<128, 128, 16, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 16, 2, 2>::LDS_BYTES)); break;
        case 27: e = launch_gemm_glds<256, 256, 16, 4, 4>(p, groups, s, occ_pad(occ, GldsCfg<256, 256, 16, 4, 4>::LDS_BYTES)); break;
        case 28: e = launch_gemm_glds<128, 64, 16, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 64, 16, 2, 2>::LDS_BYTES)); break;
        case 29: e = launch_gemm_glds<128, 64, 32, 4, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 64, 32, 4, 2>::LDS_BYTES)); break;
        case 30: e = launch_gemm_glds<128, 64, 32, 2, 2>(p, groups, s, occ_pad(occ, GldsCfg<128, 64, 32, 2, 2>::LDS_BYTES)); break;
        case 32: e = launch_gemm_glds<256, 128, 16, 4, 2, 2, true>(p, groups, s); break;  // ablation: no epilogue stores
        case 35: e = launch_gemm_glds<256, 128, 32, 4, 2, 3>(p, groups, s); break;
        case 36: e = launch_gemm_glds<64, 32, 32, 2, 1, 3>(p, groups, s); break;    // small problems: many small workgroups
        // round 6, small-M probes (configs[3]: 900 QKV tiles of 64 x 64 are 1.17 rounds of the 768 slots three 48 KB workgroups per CU give):
        // the same tile with 2 stages (32 KB: 5 per CU), with 16-deep K tiles (24 KB: 6 per CU), and both with the lean set-up + direct epilogue
        case 46: e = launch_gemm_glds<64, 64, 32, 2, 2, 2, false, 12 | 16>(p, groups, s); break;
        case 47: e = launch_gemm_glds<64, 64, 16, 2, 2, 3, false, 12 | 16>(p, groups, s); break;
        case 38: e = launch_gemm_glds<128, 32, 32, 4, 1, 3>(p, groups, s); break;
        case 39: e = launch_gemm_glds<32, 32, 32, 1, 1, 3>(p, groups, s); break;
        case 40: e = launch_gemm_glds<256, 256, 16, 4, 4, 3>(p, groups, s); break;   // one 16-wave workgroup per CU
        case 41: e = launch_gemm_glds<256, 256, 16, 4, 4, 2>(p, groups, s); break;
        case 42: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 0>(p, groups, s); break;   // t33 with workgroup barriers between epilogue slabs
        case 43: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 3>(p, groups, s); break;   // t33 + s_setprio around the MFMAs (-4 %)
        case 44: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 1>(p, groups, s); break;            // t33 with global_load_lds (64-bit per-lane pointers)
        case 45: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 5>(p, groups, s); break;            // t33 with the DMA issued right behind the barrier
        case 49: e = launch_gemm_n48<false>(p, groups, s); break;   // A/B: global_load_lds instead of buffer_load..lds   // N = 48 exactly (16x16x4 MFMA): the grouped pos-conv
        // round 4: fewer, fatter waves and the skewed schedule (OPT bit 64); plain C / R operands only (nomad_diag_gemm's are)
        case 60: e = launch_gemm_glds<256, 128, 16, 2, 2, 3, false, 13 | 16>(p, groups, s); break;        // 4 waves of 128 x 64, straight schedule
        case 61: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, true, 13 | 16>(p, groups, s); break;         // production tile 33 without its epilogue stores
        case 62: e = launch_gemm_glds<128, 128, 16, 2, 2, 3, false, 13 | 16>(p, groups, s); break;        // 4 waves of 64 x 64, 48 KB: 3 workgroups / CU
        case 63: e = launch_gemm_glds<256, 128, 16, 2, 2, 3, true, 13 | 16 | 64>(p, groups, s); break;    // tile 65 without its epilogue stores
        case 64: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64>(p, groups, s); break;   // production tile, skewed schedule
        case 65: e = launch_gemm_glds<256, 128, 16, 2, 2, 3, false, 13 | 16 | 64>(p, groups, s); break;   // 4 waves of 128 x 64, skewed schedule
        case 66: e = launch_gemm_glds<128, 128, 16, 2, 2, 3, false, 13 | 16 | 64>(p, groups, s); break;   // 4 waves of 64 x 64, skewed, 3 workgroups / CU
        case 67: e = launch_gemm_glds<128, 256, 16, 2, 2, 3, false, 13 | 16 | 64>(p, groups, s); break;   // 4 waves of 64 x 128, skewed
        case 68: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 128>(p, groups, s); break;  // production tile + per-workgroup timeline stamps (nomad_diag_timeline)
        case 69: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, true, 13 | 16 | 128>(p, groups, s); break;   // ... without the epilogue stores
        case 70: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 256>(p, groups, s); break;  // production tile, output stores paced (s_sleep 4)
        case 71: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 512>(p, groups, s); break;  // ... s_sleep 16
        case 72: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 1024>(p, groups, s); break;        // transposed accumulators + direct epilogue
        case 73: e = launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | 16 | 1024>(p, groups, s, occ_pad(occ, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)); break;
        case 74: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, true, 13 | 16 | 1024>(p, groups, s); break;         // ... without its stores
        case 75: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 1024 | 128>(p, groups, s); break;  // ... with timeline stamps
        case 76: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 128 | 2048>(p, groups, s); break;  // production tile, prologue-detail stamps
        case 77: e = launch_gemm_glds<256, 128, 16, 2, 2, 3, false, 13 | 16 | 64 | 1024>(p, groups, s); break;   // 4 waves of 128 x 64, skewed, direct epilogue
        case 78: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 4096>(p, groups, s); break;          // production tile, set-up and epilogue at raised priority
        case 79: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 1024 | 4096>(p, groups, s); break;   // direct epilogue + raised priority
        case 80: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 1024 | 4096 | 128>(p, groups, s); break;   // ... with timeline stamps
        case 81: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 4096 | 128 | 2048>(p, groups, s); break;   // production + raised priority, prologue detail
        case 86: e = launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 128 | 8192>(p, groups, s); break;   // production, set-up detail stamps
        case 14: e = launch_gemm<128, 128, 32, 2, 2, 1>(p, groups, s); break;            // ablations of tile 0
        case 15: e = launch_gemm<128, 128, 32, 2, 2, 2>(p, groups, s); break;
        case 16: e = launch_gemm<128, 128, 32, 2, 2, 3>(p, groups, s); break;
        case 17: e = launch_gemm<256, 128, 16, 4, 2, 1>(p, groups, s); break;            // ablations of tile 6
        case 18: e = launch_gemm<256, 128, 16, 4, 2, 2>(p, groups, s); break;
        case 19: e = launch_gemm<256, 128, 16, 4, 2, 3>(p, groups, s); break;
#endif
        default: return fail(NOMAD_ERR_INVALID, "gemm tile id %d is not in this library (experimental instantiations live in libnomad_diag.so)", tile);
    }
    if (e != hipSuccess) return fail(NOMAD_ERR_HIP, "gemm launch: %s", hipGetErrorString(e));
    return 0;
}

hipError_t gemm_f32_n48_split(const GemmParams& q, int groups, hipStream_t s, int S) { return launch_gemm_n48<true>(q, groups, s, S); }

// Kernel instantiation for a dense problem (measured on MI355X, tools/gemm_sweep.py, profiles/r01_gemm_sweep_*.json):
//   33 = LDS-DMA 256x128x16, 8 waves, 3-stage pipeline: best when the grid is many tiles deep (QKV, fc1, conv1-4)
//        and for the long-K / short-K N = 768 problems of the full batch (fc2, proj)
//   31 = LDS-DMA 128x128x32, 8 waves, 2-stage: out_proj, conv5/6 (N = 768 / 512 with K around 1000: 2.3 rounds of
//        256x128 tiles are too few); 34 = 128x64x32 for N not a multiple of 128
//   37 = LDS-DMA 64x64x32, 4 waves, 3-stage: small problems
// (all production instantiations (v_mfma_f32_16x16x4_f32, gemm_f32_glds_body) contract k in the same order, so the choice never changes a result bit)
// Rows of the 256 x 128 part of a two-shape launch (gemm_f32_mixed_kernel), or 0: whole rounds of the 2-per-CU workgroup slots go to
// 256 x 128 tiles, the rows of the last, partial round to 128 x 128 tiles - when that round is between 5 % and 70 % full.
// Tuning::f32_mixed = 0 switches it off; f32_mixed_m1 = <rows> (diagnostics) forces a split.
int mixed_split_rows(const nomad_ctx* c, int M, int N) {
    const Tuning& t = c->tune;
    if (!t.f32_mixed || N % 128) return 0;
    if (t.f32_mixed_m1 > 0) return t.f32_mixed_m1 < M && t.f32_mixed_m1 % 256 == 0 ? t.f32_mixed_m1 : 0;
    const int tn = N / 128, slots = t.f32_mixed_slots > 0 ? t.f32_mixed_slots : 2 * c->num_cus;
    const long long tiles = (long long)((M + 255) / 256) * tn;
    const long long rounds = tiles / slots;
    const double frac = (double)(tiles - rounds * slots) / slots;
    if (rounds < 1 || frac < t.f32_mixed_min || frac > t.f32_mixed_max) return 0;
    const long long m1 = rounds * slots / tn * 256;
    return m1 > 0 && m1 < M ? (int)m1 : 0;
}

int pick_tile(const nomad_ctx* c, int M, int N, int K) {
    const Tuning& tu = c->tune;
    const int cus = c->num_cus;
    const long long tiles256 = (long long)((M + 255) / 256) * (N / 128);
    // (NOMAD_F32_MIXED_PREFER=1, A/B: wherever the two-shape launch applies - run_gemm turns tile 33 into it - take it over the
    // 128 x 128 choice below.  Was +0.3 % of the bench step with the 32x32x2 products, is -0.4 % with 16x16x4: off.)
    if (tu.f32_mixed_prefer && N % 128 == 0 && mixed_split_rows(c, M, N) > 0) return 33;
    // 1500, not 2048: a half of the bench batch (Engine.embed runs the batch as two halves on two streams) has 1800 tiles in
    // QKV and 1600 in conv4 - the 256x128 kernel there is worth +0.5 % of the step (2230-2235 vs 2219-2222 clips/s, alternating)
    // 256 x 128 or 128 x 128 (round 4)?  Two workgroups share a CU and a lone one runs about twice as fast, so what a launch costs
    // is the largest number of tiles any CU gets: ceil(tiles / CUs) big tiles against ceil(2 tiles / CUs) half-size ones, the latter
    // ~8 % dearer per flop (more operand traffic per MFMA; 3 % before the 16x16x4 products).  conv5 at the bench batch is 1596 big tiles = 6.2 per CU -> 7, or 3192
    // small ones = 12.5 -> 13 halves = 6.5: 135 vs 127 TFLOP/s measured (profiles/r04_gemm_f32_variants.txt).  NOMAD_F32_QUANT_TILE=0:
    // the round-3 rule.
    if (tu.f32_quant_tile && N % 128 == 0 && tiles256 >= 4LL * cus) {
        const long long per_cu_256 = (tiles256 + cus - 1) / cus;
        const long long tiles128 = (long long)((M + 127) / 128) * (N / 128);
        // what a flop costs more on 128 x 128 tiles: 8 % when the host layer runs two parts of a batch concurrently (swept with the
        // 16x16x4 products: 3 / 6 / 8 / 10 / 15 % -> 2411 / 2418 / 2419 / 2418 / 2415 clips/s), 3 % for one forward at a time
        // (NOMAD_F32_QUANT_PENALTY, percent: both, A/B runs)
        const double penalty = tu.f32_quant_penalty != 0.0 ? 1.0 + tu.f32_quant_penalty / 100.0 : (tu.concurrent_parts >= 2 ? 1.08 : 1.03);
        const double cost128 = (double)((tiles128 + cus - 1) / cus) * 0.5 * penalty;
        return cost128 < (double)per_cu_256 ? 31 : 33;
    }
    if (N % 128 == 0 && tiles256 >= 1500) return 33;
    // (NOMAD_F32_LONGK_33=1, A/B: one to two rounds of 256 x 128 tiles with a long or short K - fc2 / proj of HALF a bench batch - on
    // the 256 x 128 kernel, as up to round 4; with the 16x16x4 products the 128 x 128 x 32 kernel is faster and steadier there:
    // fc2 of half a batch 133.1 against 125 TFLOP/s median, out_proj 127 against 114)
    if (tu.f32_longk_33 && N % 128 == 0 && tiles256 >= 512 && (K >= 2048 || K <= 512)) return 33;
    // less than one round of 256x128 tiles (batch 1 .. a few dozen short clips, config C4): 64x64 tiles keep the
    // most CUs busy; one wave's K loop is the latency floor there (profiles/r01_gemm_sweep_small_m.json)
    if (tiles256 < 512) return 37;
    // a few rounds of tiles with wide N (the merged training batch, M ~ 12k): 128x128 tiles, 4 waves
    // (profiles/r01_gemm_sweep_train_m.json)
    if (N >= 2048 && N % 128 == 0 && tiles256 < 2048) return 20;
    // out_proj, conv5/6 at full batch (K = 768 / 1024, N = 768 / 512): 128x128x32 tiles, 8 waves, 2 stages: +2..5 % over
    // 128x64 (profiles/r01_gemm_sweep_n768_128x128.json).  NOMAD_F32_MID_TILE=33 (A/B): the 256x128 kernel there too, so that the
    // transformer layers run ONE GEMM instantiation (no alternation)
    return N % 128 == 0 ? tu.f32_mid_tile : 34;
}

#ifdef NOMAD_DIAG
int gemm_f32_timeline_read(unsigned long long* out_host, int n) {
    HIP_TRY(hipMemcpyFromSymbol(out_host, HIP_SYMBOL(g_timeline), sizeof(unsigned long long) * 6 * (size_t)n));
    return 0;
}
#endif

extern "C" {

// ---- diagnostics -------------------------------------------------------------------------------
int nomad_diag_gemm(nomad_ctx* c, const float* A, const float* W, const float* bias, const float* R, float* C, int M,
                    int N, int K, int gelu, int tile, nomad_stream_t stream) {
    if (!c || !A || !W || !C || M <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: bad argument");
    {   // the direct epilogue (gemm_f32.hip.h OPT bit 1024) serves GEMMs without a residual only
        const int t = tile % 100;
        const bool direct_tile = t == 72 || t == 74 || t == 75 || t == 77 || t == 79 || t == 80 || t == 84 || t == 85 || t == 89 || t == 90 || t == 93 || t == 94 || t == 95;
        if (direct_tile && R) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile %d has the direct epilogue, which takes no residual", t);
    }
    // diagnostics: tile id + 100 * group_m (grouped tile order) + 10000 * occ (workgroups per CU limit)
    const int occ = tile / 10000;
    const int group_m = (tile % 10000) / 100;
    tile %= 100;
    static const int kBN[] = {128, 64, 64, 128, 128, 256, 128, 256, 256, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128, 128,
                              128, 128, 128, 128, 256, 256, 128, 256, 64, 64, 64, 128, 128, 128, 64, 128, 32, 64, 32, 32, 256, 256,
                              128, 128, 128, 128, 256, 256};
    static const int kBK[] = {32, 16, 32, 16, 32, 32, 16, 32, 16, 16, 16, 32, 32, 16, 32, 32, 32, 16, 16, 16,
                              32, 16, 16, 32, 16, 32, 16, 16, 16, 32, 32, 32, 16, 16, 32, 32, 32, 32, 32, 32, 16, 16,
                              16, 16, 16, 16, 64, 64};
    if (tile == 48 || tile == 49) {
        if (N != 48 || K % 16) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile 48 needs N == 48 and K %% 16 == 0");
        GemmParams p48 = dense(A, K, W, bias, R, C, M, N, K, gelu);
        return run_gemm(c, p48, 1, tile, static_cast<hipStream_t>(stream));
    }
    if (tile == 94 || tile == 95) {   // the shipped lean + skewed + direct-epilogue instantiation with timeline stamps (94) / set-up detail stamps (95)
        if (N % 128 || K % 32) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: N %% 128 or K %% 32 != 0");
        GemmParams pp = dense(A, K, W, bias, R, C, M, N, K, gelu);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (tile == 94) HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64 | 1024 | 16384 | 128>(pp, 1, st)));
        else HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64 | 1024 | 16384 | 128 | 8192>(pp, 1, st)));
        return 0;
    }
    if (tile == 99) {   // two tile shapes in one launch (gemm_f32_mixed_kernel): the split by mixed_split_rows (NOMAD_F32_MIXED_M1 forces one)
        if (N % 128 || K % 32 || (R && gelu)) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile 99 needs N %% 128 == 0, K %% 32 == 0, no GELU with a residual");
        GemmParams pp = dense(A, K, W, bias, R, C, M, N, K, gelu);
        hipStream_t st = static_cast<hipStream_t>(stream);
        const int m1 = mixed_split_rows(c, M, N);
        if (m1 <= 0) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile 99: no split for M = %d, N = %d", M, N);
        Scope sc(c, st, NOMAD_K_GEMM, 2.0 * M * (double)N * K, NOMAD_K_GEMM_BIG);
        constexpr int V = 13 | 16 | 64 | 16384;
        if (R) HIP_TRY((launch_gemm_mixed<V | 32768, V | 32768>(pp, m1, st)));
        else HIP_TRY((launch_gemm_mixed<V | 1024, V | 1024>(pp, m1, st)));
        return 0;
    }
    if (tile == 97 || tile == 98) {   // residual ahead (OPT bit 32768): 97 the 256 x 128 lean + skewed tile, 98 the 128 x 128 x 32 lean tile
        if (N % 128 || K % 32 || gelu || !R) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile %d needs N %% 128 == 0, K %% 32 == 0, a residual and no GELU", tile);
        GemmParams pp = dense(A, K, W, bias, R, C, M, N, K, gelu);
        hipStream_t st = static_cast<hipStream_t>(stream);
        Scope sc(c, st, NOMAD_K_GEMM, 2.0 * M * (double)N * K, NOMAD_K_GEMM_BIG);
        if (tile == 97) HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64 | 16384 | 32768>(pp, 1, st)));
        else HIP_TRY((launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | 16 | 16384 | 32768>(pp, 1, st, occ_pad(0, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES))));
        return 0;
    }
    if (tile >= 88 && tile <= 93) {   // lean set-up (OPT bit 16384) on: 88 production tile, 89 + direct epilogue, 90 + skewed + direct, 91 + skewed (LDS epilogue), 92 128x128x32 tile, 93 128x128x32 + direct
        if (N % 128 || K % 32) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: N %% 128 or K %% 32 != 0");
        GemmParams pp = dense(A, K, W, bias, R, C, M, N, K, gelu);
        hipStream_t st = static_cast<hipStream_t>(stream);
        Scope sc(c, st, NOMAD_K_GEMM, 2.0 * M * (double)N * K, NOMAD_K_GEMM_BIG);
        constexpr int L = 16384;
        switch (tile) {
            case 88: HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | L>(pp, 1, st))); break;
            case 89: HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 1024 | L>(pp, 1, st))); break;
            case 90: HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64 | 1024 | L>(pp, 1, st))); break;
            case 91: HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64 | L>(pp, 1, st))); break;
            case 92: HIP_TRY((launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | 16 | L>(pp, 1, st, occ_pad(0, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)))); break;
            default: HIP_TRY((launch_gemm_glds<128, 128, 32, 4, 2, 2, false, 12 | 16 | 1024 | L>(pp, 1, st, occ_pad(0, GldsCfg<128, 128, 32, 4, 2>::LDS_BYTES)))); break;
        }
        return 0;
    }
    if (tile == 84 || tile == 85) {   // 84: production tile, skewed schedule + direct epilogue; 85: 4 waves of 128 x 64, BK = 8 (36 KB: three workgroups / CU)
        if (N % 128 || K % 16) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: N %% 128 or K %% 16 != 0");
        GemmParams pp = dense(A, K, W, bias, R, C, M, N, K, gelu);
        Scope sc(c, static_cast<hipStream_t>(stream), NOMAD_K_GEMM, 2.0 * M * (double)N * K, NOMAD_K_GEMM_BIG);
        if (tile == 85) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile 85 (BK = 8) went with the 16x16x4 products (16-deep k groups)");
        HIP_TRY((launch_gemm_glds<256, 128, 16, 4, 2, 3, false, 13 | 16 | 64 | 1024>(pp, 1, static_cast<hipStream_t>(stream))));
        return 0;
    }
    if (tile == 82 || tile == 83 || tile == 87 || tile == 96) {   // (96: persistent, second workgroup of a CU starts half a tile late)
   // the persistent 256 x 128 kernel (83: without its output stores; 87: one workgroup per tile)
        if (N % 128 || K % 16 || K < 64) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: persistent kernel needs N %% 128 == 0, K %% 16 == 0, K >= 64");
        GemmParams pp = dense(A, K, W, bias, R, C, M, N, K, gelu);
        Scope sc(c, static_cast<hipStream_t>(stream), NOMAD_K_GEMM, 2.0 * M * (double)N * K, NOMAD_K_GEMM_BIG);
        HIP_TRY(launch_gemm_pers(pp, static_cast<hipStream_t>(stream), c->num_cus, tile == 83, tile == 87, tile == 96));
        return 0;
    }
    if (tile < 0 || (tile > 47 && (tile < 60 || tile > 81) && tile != 86)) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: tile id %d", tile);
    const int bn = (tile == 46 || tile == 47) ? 64 : tile >= 60 ? (tile == 67 ? 256 : 128) : kBN[tile], bk = tile == 46 ? 32 : tile == 47 ? 16 : tile >= 60 ? (tile == 73 ? 32 : 16) : kBK[tile];
    if (N % bn || K % bk) return fail(NOMAD_ERR_INVALID, "nomad_diag_gemm: N %% %d or K %% %d != 0", bn, bk);
    GemmParams p = dense(A, K, W, bias, R, C, M, N, K, gelu);
    p.group_m = group_m;
    return run_gemm(c, p, 1, tile, static_cast<hipStream_t>(stream), occ);
}

}  // extern "C"
