// The three phases of the transformer-stack backward — FFN sublayer, stacked-projection (q/k/v) backward with fused
// LayerNorm backward, attention core — as device functions over one 512-thread workgroup and its dynamic LDS.
// k_backward.hip wraps each in its own kernel (wide path, per-phase profiling) and chains them in k_bwd_stack: every
// phase maps DAG `dvs_bid() * NW + wave (+ gridDim.x * NW ...)` to the same wave (NW = 8 waves per workgroup, or 4 in the
// narrow mapping for small batches: dvs_api.hip, dvs_waves_per_wg), so the tiles a phase reads were written
// by the same wave in the phase before and a workgroup barrier is the only synchronisation between phases.
#pragma once
#include "dvs_backward.h"
#include "dvs_wimg.h"

constexpr int DVS_PROJB_SLOT = 6 * DVS_PKB;       // bf16 elements per wave: 3 parked tiles [hi | lo]
static size_t projb_lds_bytes(int nproj) {
    return (size_t)2 * nproj * 64 * DVS_LDB * sizeof(dvs_bf16) + 128 * 4 + (size_t)8 * DVS_PROJB_SLOT * sizeof(dvs_bf16) + 64;
}


#include "dvs_stage.h"

// Epilogue scratch (the older wave group's partial weight gradients and the vector sums, handed to the younger group through
// LDS) starts at a FIXED offset above everything a staging plan writes — images, small vectors, group counters of ANY backward
// phase kind — so that the older group can commit the next phase's plan while the younger group is still flushing this
// phase's gradients: one workgroup barrier and the whole commit (2.2-5.8 k cycles) off the critical path of every phase.
// At this point of a phase the per-wave slots that live there are dead (the DAG loop is over).
constexpr size_t DVS_BWD_EPI_FLOOR = 80 * 1024;
static_assert(8 * DVS_IMG64 * sizeof(dvs_bf16) + (192 + 64 + 128) * 4 + 16 <= DVS_BWD_EPI_FLOOR, "attention-core plan");
static_assert(7 * DVS_IMG64 * sizeof(dvs_bf16) + 6 * 64 * 4 + 16 <= DVS_BWD_EPI_FLOOR, "FFN plan");
static_assert(6 * DVS_IMG64 * sizeof(dvs_bf16) + 128 * 4 + 16 <= DVS_BWD_EPI_FLOOR, "projection plan");
// ... and ends inside the phase's own LDS footprint (matrices of 4096 floats, then [8 waves][k][64] vector partials)
static_assert(DVS_BWD_EPI_FLOOR + (2 * 4096 + 8 * 6 * 64) * 4 <= 7 * DVS_IMG64 * sizeof(dvs_bf16) + (6 * 64 + 8 * 2 * DVS_SCR) * 4, "FFN");
static_assert(DVS_BWD_EPI_FLOOR + (4096 + 8 * 64) * 4 <= 8 * DVS_IMG64 * sizeof(dvs_bf16) + (384 + 8 * 2 * DVS_SCR) * 4, "attention core");
static_assert(DVS_BWD_EPI_FLOOR + (1 * 4096 + 8 * 3 * 64) * 4 <= 2 * DVS_IMG64 * sizeof(dvs_bf16) + 8 * DVS_PROJB_SLOT * sizeof(dvs_bf16), "1 projection");
static_assert(DVS_BWD_EPI_FLOOR + (3 * 4096 + 8 * 5 * 64) * 4 <= 6 * DVS_IMG64 * sizeof(dvs_bf16) + 8 * DVS_PROJB_SLOT * sizeof(dvs_bf16), "3 projections");
__device__ __forceinline__ float* dvs_bwd_epi(char* smem) { return (float*)(smem + DVS_BWD_EPI_FLOOR); }

#ifdef DVS_STAMPS
DVS_STAMP_DECL(dvs_stamps_bwd);
#endif

#ifndef DVS_TOUCH_MODE
#define DVS_TOUCH_MODE 2
#endif
// -DDVS_TOUCH_ROUND (experiment): inside the DAG loop every wave touches the cold tile of its NEXT round at the top of the
// current one (one load instruction, dvs_touch_first's idea applied within a phase)
#ifdef DVS_TOUCH_ROUND
#define DVS_ROUND_TOUCH(name, buf, tile_now, tile_next, ntiles, L) \
    const float name = (buf)[(size_t)((tile_next) < (ntiles) ? (tile_next) : (tile_now)) * 1024 + (size_t)(L).lane * 16]
#define DVS_ROUND_TOUCH_DONE(name) asm volatile("" ::"v"(name))
#else
#define DVS_ROUND_TOUCH(name, buf, tile_now, tile_next, ntiles, L) ((void)0)
#define DVS_ROUND_TOUCH_DONE(name) ((void)0)
#endif

// ---- LDS layouts of the three phase kinds ----------------------------------------------------------------------------
struct FfnBLds {
    // bf16x3 images (dvs_bf16.h) of W2^T and W1^T (d hidden, d x); W1 as the bf16x6 triple k_ffn_fwd uses: the hidden is
    // recomputed with the forward's own instruction sequence, because its sign must reproduce the forward's ReLU mask.
    // The weight gradients run on the bf16 pipe too (dvs_coop_dw_bf: parked bf16 tiles, transposing LDS reads).
    dvs_bf16 *W2Th, *W2Tl, *W1Th, *W1Tl, *W1x6;
    float *b1, *b2, *lg, *lb, *og, *ob, *slots;
    int* gcount;
};
DVS_HD inline FfnBLds ffnb_lds(char* smem) {
    FfnBLds l;
    l.W2Th = (dvs_bf16*)smem;
    l.W2Tl = l.W2Th + 64 * DVS_LDB;
    l.W1Th = l.W2Tl + 64 * DVS_LDB;
    l.W1Tl = l.W1Th + 64 * DVS_LDB;
    l.W1x6 = l.W1Tl + 64 * DVS_LDB;
    l.b1 = (float*)(l.W1x6 + 3 * 64 * DVS_LDB);
    l.b2 = l.b1 + 64;
    l.lg = l.b2 + 64;
    l.lb = l.lg + 64;
    l.og = l.lb + 64;
    l.ob = l.og + 64;
    l.gcount = (int*)(l.ob + 64);              // group-barrier counters: below DVS_BWD_EPI_FLOOR in every layout (see there)
    l.slots = (float*)(l.gcount + 4);
    return l;
}
static size_t ffnb_lds_bytes() {
    return 7 * 64 * DVS_LDB * sizeof(dvs_bf16) + (6 * 64 + (size_t)8 * 2 * DVS_SCR + 16) * sizeof(float);
}

constexpr int DVS_ATTNB_TLD = 20;             // floats per row of the per-wave transposing tile (16 + 4: 16-byte rows, 2-way banks at most)
constexpr int DVS_ATTNB_TSCR = 16 * DVS_ATTNB_TLD;
struct AttnBLds {
    float *inb, *outb, *lg, *lb, *slots, *stats;
    // bf16x3 images (dvs_bf16.h): the in-projection rows (q, k, v are recomputed through a softmax — smooth, so their
    // ~1e-5 perturbation stays a ~1e-5 perturbation of the gradient; the FFN's hidden, whose SIGN is a mask, is recomputed
    // with the forward's own bf16x6 sequence instead) and Wo^T (dO^T = Wo^T dy^T, a pure gradient product)
    dvs_bf16 *Winh, *Winl, *WoTh, *WoTl;
    int* gcount;
};
DVS_HD inline AttnBLds attnb_lds(char* smem) {
    AttnBLds l;
    l.WoTh = (dvs_bf16*)smem;                  // global order of the block (dvs_wimg.h): WoutT pair, WinB pair
    l.WoTl = l.WoTh + 64 * DVS_LDB;
    l.Winh = l.WoTl + 64 * DVS_LDB;
    l.Winl = l.Winh + 192 * DVS_LDB;
    l.inb = (float*)(l.Winl + 192 * DVS_LDB);
    l.outb = l.inb + 192;
    l.lg = l.outb + 64;
    l.lb = l.lg + 64;
    l.gcount = (int*)(l.lb + 64);
    l.slots = (float*)(l.gcount + 4);          // per wave: A (d y, row-major) and B (transpose scratch, then O)
    l.stats = l.slots + 8 * 2 * DVS_SCR;       // per wave DVS_ATTNB_TSCR floats: transposing tile of the core
    return l;
}
static size_t attnb_lds_floats() {
    return 512 * DVS_LDB / 2 + 192 + 64 + 128 + (size_t)8 * 2 * DVS_SCR + 8 * DVS_ATTNB_TSCR + 16;
}


struct ProjBLds {
    dvs_bf16* WT;                // [NPROJ][hi | lo][64][LDB]
    float *lg, *lb;
    dvs_bf16* slots;             // per wave 3 tiles [hi | lo]: A0, A1 (alternating dY) and B (X)
    int* gcount;
};
DVS_HD inline ProjBLds projb_lds(char* smem, int nproj) {
    ProjBLds l;
    l.WT = (dvs_bf16*)smem;
    l.lg = (float*)(l.WT + nproj * 2 * DVS_IMG64);
    l.lb = l.lg + 64;
    l.gcount = (int*)(l.lb + 64);
    l.slots = (dvs_bf16*)(l.gcount + 4);
    return l;
}

// ---- staging plans (dvs_stage.h): what each phase kind keeps in LDS --------------------------------------------------
inline void ffnb_plan(DvsStagePlan& p, const FfnBwdArgs& a, char* smem) {
    const FfnBLds l = ffnb_lds(smem);
    dvs_plan_clear(p);
    dvs_plan_seg(p, smem, l.W2Th, (const dvs_bf16*)a.wimg + DvsFfnImg::W2T, (int)(7 * DVS_IMG64));   // W2^T, W1^T x3 pairs, W1 x6
    dvs_plan_vec(p, smem, l.b1, a.l1_b, 64);
    dvs_plan_vec(p, smem, l.b2, a.l2_b, 64);
    dvs_plan_vec(p, smem, l.lg, a.ln.g, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.lb, a.ln.b, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.og, a.own.g, a.own_pre ? 64 : 0);
    dvs_plan_vec(p, smem, l.ob, a.own.b, a.own_pre ? 64 : 0);
    p.zero_int = (int)(((const char*)l.gcount - smem) >> 2);
    dvs_plan_cold(p, a.xin, a.own_pre, a.dims.B * a.dims.NT);
    dvs_plan_seal(p);
}
inline void attnb_plan(DvsStagePlan& p, const AttnBwdArgs& a, char* smem) {
    const AttnBLds l = attnb_lds(smem);
    dvs_plan_clear(p);
    dvs_plan_seg(p, smem, l.WoTh, (const dvs_bf16*)a.wimg + DvsAttnImg::WoutT, (int)(8 * DVS_IMG64));   // Wo^T pair, then parts hi, mid of Win
    dvs_plan_vec(p, smem, l.inb, a.in_b, 192, true);
    dvs_plan_vec(p, smem, l.outb, a.out_b, 64);
    dvs_plan_vec(p, smem, l.lg, a.ln.g, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.lb, a.ln.b, a.ln.stats ? 64 : 0);
    p.zero_int = (int)(((const char*)l.gcount - smem) >> 2);
    dvs_plan_cold(p, a.xin, a.kv, a.dims.B * a.dims.NT);
    dvs_plan_seal(p);
}
inline void projb_plan(DvsStagePlan& p, const ProjBwdArgs& a, int nproj, char* smem) {
    const ProjBLds l = projb_lds(smem, nproj);
    dvs_plan_clear(p);
    dvs_plan_seg(p, smem, l.WT, a.wimg, (int)(nproj * 2 * DVS_IMG64));
    dvs_plan_vec(p, smem, l.lg, a.ln.g, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.lb, a.ln.b, a.ln.stats ? 64 : 0);
    p.zero_int = (int)(((const char*)l.gcount - smem) >> 2);
    dvs_plan_cold(p, a.xin, a.xin2, a.dims.B * a.dims.NT);
    dvs_plan_seal(p);
}
inline void dvs_bwd_plan(DvsStagePlan& p, const BwdPhase& ph, char* smem) {
    switch (ph.kind) {
        case DVS_PH_FFN: ffnb_plan(p, ph.u.f, smem); break;
        case DVS_PH_ATTN: attnb_plan(p, ph.u.a, smem); break;
        case DVS_PH_PROJ1: projb_plan(p, ph.u.p, 1, smem); break;
        case DVS_PH_PROJ2: projb_plan(p, ph.u.p, 2, smem); break;
        default: projb_plan(p, ph.u.p, 3, smem); break;
    }
}
// Tail of every phase: the plan of the phase that follows in the chained launch (null: none) is fetched while this one ends.
struct DvsBwdTail {
    DvsPrefetch<DVS_PF_BWD_TAIL> pf;
};
template <class PP>
__device__ __forceinline__ void dvs_tail_issue(DvsBwdTail& t, PP next, bool has_next) {
    if (has_next && dvs_tid() < DVS_PF_THREADS) dvs_prefetch_issue<false>(t.pf, next, dvs_tid(), DVS_PF_THREADS);
}
// call after the last LDS access of the epilogue, with a workgroup barrier in between; the caller's next barrier publishes it
template <class PP>
__device__ __forceinline__ void dvs_tail_commit(const DvsBwdTail& t, PP next, bool has_next, char* smem) {
    if (has_next && dvs_tid() < DVS_PF_THREADS) dvs_prefetch_commit<false>(t.pf, next, smem, dvs_tid(), DVS_PF_THREADS);
}

// ---------------------------------------------------------------------------------------------------------
// FFN sublayer backward (autograd of pace.py:62-65 / 151-153).  Recomputes h = drop(relu(W1 x + b1)) from the saved
// pre-sum of the producing sublayer; all four products (dW2, dh, dW1, dx) are MFMA chains on registers.
// ---------------------------------------------------------------------------------------------------------
// 8 waves per workgroup, one DAG per wave per iteration; weight gradients are accumulated cooperatively
// (dvs_coop_dw_bf): ~150 registers per lane, two waves per SIMD, so one wave's VALU phases overlap the other's MFMAs.
// The gradient products (d hidden, d x, and the weight gradients) run on the bf16 matrix pipe as bf16x3: gradient
// parity is bounded at 2e-3 of the tensor maximum (tests), two orders of magnitude above their ~1e-5 error, whereas
// the forward keeps fp32-accurate bf16x6 products for the 1e-4 ELBO contract and the hidden is recomputed with them.
// mine / stage_mine: this phase's staging plan; stage_mine is false when the previous phase of the chain already put it into
// LDS (and a barrier has passed since).  next / has_next: the plan of the phase that follows, fetched in this phase's tail
// (dvs_stage.h).  PP: plan pointer type (dvs_stage.h: plain, or into the kernel-argument segment).
template <int NW, class PP>
__device__ __forceinline__ void dvs_ffn_bwd_phase(const FfnBwdArgs& a, char* smem, PP mine, bool stage_mine, PP next,
                                                  bool has_next) {
    const FfnBLds l = ffnb_lds(smem);
    DVS_STAMP(dvs_stamps_bwd, mine, 0);
    if (stage_mine) {
        dvs_stage_now<(NW == 8 ? DVS_PF_BWD : DVS_PF_BWD_TAIL)>(mine, smem);
        __syncthreads();
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 1);
    int* gcount = l.gcount;
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int B = a.dims.B * a.dims.NT;                    // tiles (dvs_tile_of): the sublayer is token-local
    float* sA = l.slots + L.wave * 2 * DVS_SCR;
    float* sB = sA + DVS_SCR;
    DvsGroup G = {gcount + (L.wave >> 2), 0};
    f4 aW1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()}, aW2[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
    f4 ab1 = f4_zero(), ab2 = f4_zero();                                // bias gradients: rows 16*(wave&3).. like aW1 / aW2
    float vgam = 0.f, vbet = 0.f, vog = 0.f, vob = 0.f;                   // lane = feature
    dvs_bf16* const bslots = (dvs_bf16*)l.slots;                        // a [hi | lo] bf16 pair fills one fp32 scratch tile
    constexpr int BSTRIDE = 2 * 2 * DVS_SCR;                            // bf16 elements between the slots of two waves
    bool gate = !stage_mine;                               // chained: the barrier that publishes this phase's images (DVS_PHASE_GATE)
    DVS_PHASE_GATE_INIT(gate);
    dvs_stagger(L.wave);
    for (int base = dvs_bid() * NW; base < B; base += gridDim.x * NW) {
        const int dag = base + L.wave;                     // tile index
        const bool live = dag < B;
        const size_t dg = live ? dag : 0;
        const DvsTile T = dvs_tile_of((int)dg, a.dims);
        const int N = T.Nl;
        const int Nl = live ? N : 0;                       // a wave without a tile carries all-zero tiles
        f4 x[4], xhat[4], gp[4];
        float rstd;
        DVS_ROUND_TOUCH(rt, a.xin, dg, dg + gridDim.x * NW, B, L);
        {
            DvsRawX rx;
            dvs_load_x_issue(rx, a.xin, a.ln, dg, L);      // the COLD load of the round (a saved forward activation; the gradient
            DVS_PHASE_GATE(gate);                          // tile below was written by this wave one phase ago); gate: behind it,
            dvs_load_tile(gp, a.gpre, dg, L);              // ahead of the first LDS access.  More tiles in flight across the gate
            dvs_load_x_finish<true>(x, xhat, rstd, rx, a.ln, l.lg, l.lb, Nl, L);      // cost the 8-wave chain 14 VGPR spills
            dvs_zero_rows(gp, Nl, L);
        }
        if (a.own_pre) {   // incoming gradient is w.r.t. LN_own(pre_own): pull back to d(pre_own)
            f4 po[4], pxh[4], t0[4];
            float prstd;
            dvs_load_x<true>(po, pxh, prstd, a.own_pre, a.own, l.og, l.ob, dg, Nl, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) t0[t] = gp[t] * pxh[t];
            dvs_park_T(sA, t0, L);
            dvs_park_T(sB, gp, L);
            dvs_wave_sync();
            vog += dvs_colsum(sA, L);
            vob += dvs_colsum(sB, L);
            dvs_wave_sync();
            dvs_ln_bwd_core(gp, pxh, prstd, l.og, L);
        }
        const uint32_t gdag = a.dims.dag_offset + (uint32_t)T.dag;
        const uint32_t khid = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_hidden, gdag);
        const uint32_t kpost = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag);
        // recompute hidden
        f4 hpre[4], hd[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) hpre[t] = dvs_vecT(l.b1, t, L);
        dvs_matb3<4>(hpre, dvs_split3_T(x), l.W1x6, 64, 0, L);   // the forward's own bf16x6 product, bit for bit: its sign is the ReLU mask
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) hd[t][kk] = (live && L.r < N) ? fmaxf(hpre[t][kk], 0.f) : 0.f;
        const uint32_t hid_bits = dvs_dropout_bits(khid, D, L, T.tok0);      // drawn once, applied to h and to d h
        dvs_dropout_apply(hd, hid_bits, D);
        // dy = d(W2 h + b2) = dropout-mask(post) applied to d pre
        f4 dy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dy[t] = gp[t];
        dvs_dropout_tile(dy, kpost, D, L, T.tok0);
        // ---- dW2 += dy^T hd, db2 += sum dy --------------------------------------------------------------------------
        dvs_park_bf((dvs_bf16*)sA, dy, L);
        dvs_park_bf((dvs_bf16*)sB, hd, L);
        dvs_group_barrier(G, L);
        dvs_coop_dw_bf(aW2, ab2, bslots, bslots + 2 * DVS_SCR, BSTRIDE, L);
        f4 dh[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
        dvs_matb_T<4>(dh, dvs_split_T(dy), l.W2Th, l.W2Tl, 0, L);
        dvs_dropout_apply(dh, hid_bits, D);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dh[t][kk] = hpre[t][kk] > 0.f ? dh[t][kk] : 0.f;
        dvs_group_barrier(G, L);
        // ---- dW1 += dh^T x, db1 += sum dh ----------------------------------------------------------------------------
        dvs_park_bf((dvs_bf16*)sA, dh, L);
        dvs_park_bf((dvs_bf16*)sB, x, L);
        dvs_group_barrier(G, L);
        dvs_coop_dw_bf(aW1, ab1, bslots, bslots + 2 * DVS_SCR, BSTRIDE, L);
        f4 dx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dx[t] = gp[t];
        dvs_matb_T<4>(dx, dvs_split_T(dh), l.W1Th, l.W1Tl, 0, L);
        dvs_group_barrier(G, L);
        if (a.ln.stats) {
            f4 t0[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) t0[t] = dx[t] * xhat[t];
            dvs_park_T(sA, t0, L);
            dvs_park_T(sB, dx, L);
            dvs_wave_sync();
            vgam += dvs_colsum(sA, L);
            vbet += dvs_colsum(sB, L);
            dvs_wave_sync();
            dvs_ln_bwd_core(dx, xhat, rstd, l.lg, L);
        }
        if (live) dvs_store_tile(a.gout, dag, dx, L);
        DVS_ROUND_TOUCH_DONE(rt);
        if (base == dvs_bid() * NW) DVS_STAMP(dvs_stamps_bwd, mine, 7);      // end of the first round (tools/phase_stamps.py)
    }
    DVS_PHASE_GATE(gate);                                  // a workgroup without a tile
    DVS_STAMP(dvs_stamps_bwd, mine, 2);
    // every wave touches its first tiles of the next phase (dvs_stage.h) — BEHIND the older group's image prefetch: ahead of it
    // the cold reads delay the issue of the 73 KB of image loads, which for the short phases sits on the critical path
    // (DVS_TOUCH_MODE: 0 none, 1 ahead of the prefetch, 2 behind it; A/B builds)
#if DVS_TOUCH_MODE == 1
    const DvsTouch touch = dvs_touch_first(next, has_next, NW, L.wave, L.lane);
#endif
    DvsBwdTail tail;
    dvs_tail_issue(tail, next, has_next);
#if DVS_TOUCH_MODE == 2
    const DvsTouch touch = dvs_touch_first(next, has_next, NW, L.wave, L.lane);
#elif DVS_TOUCH_MODE == 0
    const DvsTouch touch = {{0.f, 0.f}};
#endif
    DVS_STAMP(dvs_stamps_bwd, mine, 3);
    dvs_lds_barrier();
    DVS_STAMP(dvs_stamps_bwd, mine, 4);
    float* slab = a.slab + (size_t)dvs_bid() * a.P;
    float* const buf1 = dvs_bwd_epi(smem);             // one 64 x 64 staging buffer per matrix, then the vector sums
    float* const buf2 = buf1 + 4096;
    float* red = buf2 + 4096;                         // [8 waves][6][64]
    dvs_coop_stage<NW>(buf1, aW1, L);
    dvs_coop_stage<NW>(buf2, aW2, L);
    red[(L.wave * 6 + 0) * 64 + L.lane] = 0.f;
    red[(L.wave * 6 + 1) * 64 + L.lane] = 0.f;
    red[(L.wave * 6 + 2) * 64 + L.lane] = vgam;
    red[(L.wave * 6 + 3) * 64 + L.lane] = vbet;
    red[(L.wave * 6 + 4) * 64 + L.lane] = vog;
    red[(L.wave * 6 + 5) * 64 + L.lane] = vob;
    dvs_wave_sync();
    if (L.r == 0) {                                   // this wave's 16 features of the bias gradients (dvs_coop_dw_bf)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            red[(L.wave * 6 + 0) * 64 + 16 * (L.wave & 3) + 4 * L.g + reg] = ab1[reg];
            red[(L.wave * 6 + 1) * 64 + 16 * (L.wave & 3) + 4 * L.g + reg] = ab2[reg];
        }
    }
    dvs_lds_barrier();
    dvs_coop_flush<NW>(buf1, slab + a.o_l1_w, aW1, L);
    dvs_coop_flush<NW>(buf2, slab + a.o_l2_w, aW2, L);
    for (int k = dvs_tid() >> 6; k < 6; k += NW) {          // one wave per vector (8 waves: the first six, one pass)
        const int f = dvs_tid() & 63;
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += red[(w * 6 + k) * 64 + f];
        const int64_t off = k == 0 ? a.o_l1_b : k == 1 ? a.o_l2_b : k == 2 ? a.o_ln_g : k == 3 ? a.o_ln_b : k == 4 ? a.o_own_g : a.o_own_b;
        if (off >= 0) slab[off + f] = s;
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 5);
    dvs_tail_commit(tail, next, has_next, smem);       // below DVS_BWD_EPI_FLOOR: does not touch what the flush still reads
    DVS_STAMP(dvs_stamps_bwd, mine, 6);
    dvs_touch_done(touch);
}

// ---------------------------------------------------------------------------------------------------------
// Backward of NPROJ stacked 64->64 projections of one input X (the q/k/v in-projections of nn.MultiheadAttention):
//   dX^T = sum_p W_p^T dY_p^T (+ residual) ; dW_p += dY_p(N) (x) X(N) ; db_p += sum_tok dY_p ; then the producing
//   sublayer's LayerNorm backward.  Used for self-attention (NPROJ=3), cross-attention q (1) and k,v (2, X = memory).
// ---------------------------------------------------------------------------------------------------------
// 8 waves per workgroup in two independent groups of four; weight gradients accumulated cooperatively (dvs_coop_dw_bf):
// per wave 16 accumulator registers per projection, ~130 VGPRs, two waves per SIMD.  LDS slots per wave: X (kept for
// all projections of the DAG) and two alternating dY slots, so one group barrier per projection + one per DAG.
template <int NPROJ, int NW, class PP>
__device__ __forceinline__ void dvs_proj_bwd_phase(const ProjBwdArgs& a, char* smem, PP mine, bool stage_mine, PP next,
                                                   bool has_next) {
    // W_p^T as bf16x3 images (dvs_bf16.h): dX^T = sum_p W_p^T dY_p^T is a pure gradient product (no mask or statistic
    // of the forward depends on it), so it runs on the bf16 matrix pipe.  So do the weight gradients dW_p = dY_p^T X: the
    // tiles are parked as bf16 hi / lo images and read back transposed (dvs_coop_dw_bf); the bias gradients ride along as
    // products with a ones fragment.
    const ProjBLds pl = projb_lds(smem, NPROJ);
    dvs_bf16* WT = pl.WT;
    float* lg = pl.lg;
    float* lb = pl.lb;
    dvs_bf16* slots = pl.slots;
    int* gcount = pl.gcount;
    DVS_STAMP(dvs_stamps_bwd, mine, 0);
    if (stage_mine) {
        dvs_stage_now<(NW == 8 ? DVS_PF_BWD : DVS_PF_BWD_TAIL)>(mine, smem);
        __syncthreads();
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 1);
    const Lane L = dvs_lane();
    const int B = a.dims.B * a.dims.NT;              // tiles
    dvs_bf16* myA0 = slots + L.wave * DVS_PROJB_SLOT;
    dvs_bf16* myA1 = myA0 + 2 * DVS_PKB;
    dvs_bf16* myB = myA0 + 4 * DVS_PKB;
    DvsGroup G = {gcount + (L.wave >> 2), 0};
    bool gate = !stage_mine;                               // chained: the barrier that publishes this phase's images (DVS_PHASE_GATE)
    DVS_PHASE_GATE_INIT(gate);
    f4 aW[NPROJ][4], ab[NPROJ];
    float vgam = 0.f, vbet = 0.f;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) {
        ab[p] = f4_zero();
#pragma unroll
        for (int i = 0; i < 4; ++i) aW[p][i] = f4_zero();
    }
    dvs_stagger(L.wave);
    for (int base = dvs_bid() * NW; base < B; base += gridDim.x * NW) {
        const int dag = base + L.wave;               // tile index
        const bool live = dag < B;
        const size_t dg = live ? dag : 0;
        const int Nl = live ? dvs_tile_of((int)dg, a.dims).Nl : 0;
        f4 x[4], xhat[4], dx[4];
        float rstd;
        DVS_ROUND_TOUCH(rt, a.xin, dg, dg + gridDim.x * NW, B, L);
        {
            DvsRawX rx;
            dvs_load_x_issue(rx, a.xin, a.ln, dg, L);      // the cold load of the round
            DVS_PHASE_GATE(gate);                          // behind it, ahead of the first LDS access
            if (a.gres) {
                dvs_load_tile(dx, a.gres, dg, L);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) dx[t] = f4_zero();
            }
            dvs_load_x_finish<true>(x, xhat, rstd, rx, a.ln, lg, lb, Nl, L);
            dvs_zero_rows(dx, Nl, L);
        }
        dvs_park_bf(myB, x, L);
        const bool split = NPROJ == 3 && a.xin2 != nullptr;      // uniform: q of X, then k and v of X2 (dvs_backward.h)
        // d X += W_p^T dY_p, dW_p += dY_p^T X over projections p0 .. p1 - 1 against the tile parked in myB
        auto run = [&](int p0, int p1, f4 (&acc_dx)[4]) {
#pragma unroll
            for (int p = 0; p < NPROJ; ++p) {
                if (p < p0 || p >= p1) continue;
                dvs_bf16* mine = (p & 1) ? myA1 : myA0;
                f4 dy[4];
                dvs_load_grad(dy, a.gy[p], dg, Nl, L);
                dvs_park_bf(mine, dy, L);
                dvs_group_barrier(G, L);
                dvs_coop_dw_bf(aW[p], ab[p], slots + (p & 1) * 2 * DVS_PKB, slots + 4 * DVS_PKB, DVS_PROJB_SLOT, L);
                dvs_matb_T<4>(acc_dx, dvs_split_T(dy), WT + p * 2 * DVS_IMG64, WT + p * 2 * DVS_IMG64 + DVS_IMG64, 0, L);
            }
        };
        run(0, split ? 1 : NPROJ, dx);
        dvs_group_barrier(G, L);        // every wave of the group is done with this DAG's slots
        if (a.ln.stats) {
            // two fp32 tiles for the column sums: each fits the [hi | lo] pair of a parked tile
            float* fA = (float*)myA0;
            float* fB = (float*)myB;
            f4 t0[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) t0[t] = dx[t] * xhat[t];
            dvs_park_T(fA, t0, L);
            dvs_park_T(fB, dx, L);
            dvs_wave_sync();
            vgam += dvs_colsum(fA, L);
            vbet += dvs_colsum(fB, L);
            dvs_wave_sync();
            dvs_ln_bwd_core(dx, xhat, rstd, lg, L);
        }
        if (live) {
            if (a.accumulate_out) {
                f4 old[4];
                dvs_load_tile(old, a.gout, dag, L);
#pragma unroll
                for (int t = 0; t < 4; ++t) dx[t] += old[t];
            }
            dvs_store_tile(a.gout, dag, dx, L);
        }
        if (split) {
            // second input (the decoder memory: no LayerNorm), projections 1 and 2; its gradient accumulates over the layers
            f4 x2[4], dx2[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
            dvs_load_grad(x2, a.xin2, dg, Nl, L);
            dvs_park_bf(myB, x2, L);
            run(1, 3, dx2);
            dvs_group_barrier(G, L);
            if (live) {
                if (a.accumulate_out2) {
                    f4 old[4];
                    dvs_load_tile(old, a.gout2, dag, L);
#pragma unroll
                    for (int t = 0; t < 4; ++t) dx2[t] += old[t];
                }
                dvs_store_tile(a.gout2, dag, dx2, L);
            }
        }
        DVS_ROUND_TOUCH_DONE(rt);
        if (base == dvs_bid() * NW) DVS_STAMP(dvs_stamps_bwd, mine, 7);
    }
    DVS_PHASE_GATE(gate);                                  // a workgroup without a tile
    DVS_STAMP(dvs_stamps_bwd, mine, 2);
    // every wave touches its first tiles of the next phase (dvs_stage.h) — BEHIND the older group's image prefetch: ahead of it
    // the cold reads delay the issue of the 73 KB of image loads, which for the short phases sits on the critical path
    // (DVS_TOUCH_MODE: 0 none, 1 ahead of the prefetch, 2 behind it; A/B builds)
#if DVS_TOUCH_MODE == 1
    const DvsTouch touch = dvs_touch_first(next, has_next, NW, L.wave, L.lane);
#endif
    DvsBwdTail tail;
    dvs_tail_issue(tail, next, has_next);
#if DVS_TOUCH_MODE == 2
    const DvsTouch touch = dvs_touch_first(next, has_next, NW, L.wave, L.lane);
#elif DVS_TOUCH_MODE == 0
    const DvsTouch touch = {{0.f, 0.f}};
#endif
    DVS_STAMP(dvs_stamps_bwd, mine, 3);
    dvs_lds_barrier();
    DVS_STAMP(dvs_stamps_bwd, mine, 4);
    float* slab = a.slab + (size_t)dvs_bid() * a.P;
    const bool so = a.slot_order != 0;
    float* const bufs = dvs_bwd_epi(smem);             // NPROJ staging buffers of 4096 floats, then the vector sums
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) dvs_coop_stage<NW>(bufs + 4096 * p, aW[p], L);
    float* red = bufs + 4096 * NPROJ;                 // [8 waves][NPROJ + 2][64]
    // bias gradients: wave (group, ot) holds the sums of features 16*ot + 4g + reg (every column r the same)
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) red[(L.wave * (NPROJ + 2) + p) * 64 + L.lane] = 0.f;
    red[(L.wave * (NPROJ + 2) + NPROJ) * 64 + L.lane] = vgam;
    red[(L.wave * (NPROJ + 2) + NPROJ + 1) * 64 + L.lane] = vbet;
    dvs_wave_sync();
    if (L.r == 0) {
#pragma unroll
        for (int p = 0; p < NPROJ; ++p)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) red[(L.wave * (NPROJ + 2) + p) * 64 + 16 * (L.wave & 3) + 4 * L.g + reg] = ab[p][reg];
    }
    dvs_lds_barrier();
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) dvs_coop_flush<NW>(bufs + 4096 * p, slab + a.o_w + 4096 * p, aW[p], L, so, false);
    for (int k = dvs_tid() >> 6; k < NPROJ + 2; k += NW) {
        const int f = dvs_tid() & 63;
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += red[(w * (NPROJ + 2) + k) * 64 + f];
        if (k < NPROJ) slab[a.o_b + 64 * k + (so ? dvs_pi(f) : f)] = s;
        else if (a.o_ln_g >= 0) slab[(k == NPROJ ? a.o_ln_g : a.o_ln_b) + f] = s;
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 5);
    dvs_tail_commit(tail, next, has_next, smem);
    DVS_STAMP(dvs_stamps_bwd, mine, 6);
    dvs_touch_done(touch);
}

// ---------------------------------------------------------------------------------------------------------
// Attention-core backward: from d(pre) of an attention sublayer to d(q), d(k), d(v) projections and dWo/dbo.
//
// Recomputes q,k,v and the probabilities of every head (nothing but the sublayer input was saved).  Two
// orientations of the 16x16 score tile are used so that every product contracts over the MFMA row index of a
// register-resident operand (see dvs_device.h):
//   "T":  P^T[j=4g+reg][i=r]  -> softmax statistics (in-lane + 2 shuffles), dP^T, dS^T -> dq^T
//   "S":  P  [i=4g+reg][j=r]  -> recomputed from the T statistics (3 shuffles), dP, dS -> dk^T, dv^T
// q,k (T-layout) feed the score products directly; their N-layout copies (for dq/dk), v^T and dO (N) come from
// per-wave LDS transposes.

// Dropout keep-bits of the attention probabilities of one DAG, all 8 heads.  T orientation: lane (r, g) owns the elements
// (query i = r, key j = 4g + reg) -> bit 4h + reg of `T`, drawn exactly like the forward does (element index
// (h*16 + i)*16 + j, two elements per draw).  S orientation: lane (r, g) owns (i = 4g + reg, j = r), i.e. the bits that
// lane (r' = 4g + reg, g' = r >> 2) holds at reg' = r & 3: four cross-lane reads instead of 32 more hash draws.
// v if bit `pos` of `bits` is set, else +0: a signed 1-bit field extract (0 / -1) ANDed onto the value
__device__ __forceinline__ float dvs_bit_select(uint32_t bits, int pos, float v) {
#ifndef DVS_EMU
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)bits, pos, 1);      // v_bfe_i32: 0 or -1
#else
    const uint32_t m = ((bits >> pos) & 1u) ? 0xFFFFFFFFu : 0u;
#endif
    return __uint_as_float(__float_as_uint(v) & m);
}
struct ProbMask {
    uint32_t T;
    uint32_t S[4];      // S[reg] >> (4h) & 1: element (i = 4g + reg, j = r) of head h
};
__device__ __forceinline__ ProbMask dvs_prob_mask(uint32_t key, const DvsDrop& D, const Lane& L) {
    ProbMask m;
    m.T = 0xFFFFFFFFu;
    if (D.on) {
        uint32_t bits = 0;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const uint32_t p0 = (uint32_t)((h * 16 + L.r) * 8 + 2 * L.g);
            const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
            bits |= ((h0 & 0xFFFFu) >= D.thr16 ? 1u : 0u) << (4 * h);
            bits |= ((h0 >> 16) >= D.thr16 ? 1u : 0u) << (4 * h + 1);
            bits |= ((h1 & 0xFFFFu) >= D.thr16 ? 1u : 0u) << (4 * h + 2);
            bits |= ((h1 >> 16) >= D.thr16 ? 1u : 0u) << (4 * h + 3);
        }
        m.T = bits;
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg)
        m.S[reg] = D.on ? ((uint32_t)__shfl((int)m.T, 4 * L.g + reg + 16 * (L.r >> 2)) >> (L.r & 3)) : 0xFFFFFFFFu;
    return m;
}
__device__ __forceinline__ f4 mask_T(const ProbMask& m, int h, const DvsDrop& D) {
    if (!D.on) return f4_splat(1.f);
    f4 r;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) r[reg] = dvs_bit_select(m.T, 4 * h + reg, D.scale);
    return r;
}
__device__ __forceinline__ f4 mask_S(const ProbMask& m, int h, const DvsDrop& D) {
    if (!D.on) return f4_splat(1.f);
    f4 r;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) r[reg] = dvs_bit_select(m.S[reg], 4 * h, D.scale);
    return r;
}

// 8 waves per workgroup in two independent groups of four (dvs_backward.h); one DAG per wave per iteration.  The
// out-projection gradient is accumulated cooperatively from the parked d y and O tiles, d q / d k / d v tiles are stored
// as soon as their head pair is finished, so a wave stays within 256 registers and two waves share each SIMD.
template <int NW, class PP>
__device__ __forceinline__ void dvs_attn_bwd_phase(const AttnBwdArgs& a, char* smem, PP mine, bool stage_mine, PP next,
                                                   bool has_next) {
    const AttnBLds l = attnb_lds(smem);
    DVS_STAMP(dvs_stamps_bwd, mine, 0);
    if (stage_mine) {
        dvs_stage_now<(NW == 8 ? DVS_PF_BWD : DVS_PF_BWD_TAIL)>(mine, smem);
        __syncthreads();
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 1);
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N, B = a.dims.B;
    float* sA = l.slots + L.wave * 2 * DVS_SCR;
    float* sB = sA + DVS_SCR;
    float* st = l.stats + L.wave * DVS_ATTNB_TSCR;
    DvsGroup G = {l.gcount + (L.wave >> 2), 0};
    const float scale = 0.35355339059327373f;
    f4 aWo[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
    f4 abo = f4_zero();                           // d out_proj.bias, rows 16*(wave&3).. like aWo
    // chained: the barrier that publishes this phase's images — taken up front here; inside the DAG loop (behind the first
    // round's loads, as the FFN / projection phases do) it cost this phase, which sits at the 256-register limit, 15 spills
    bool gate = !stage_mine;
    DVS_PHASE_GATE(gate);
    dvs_stagger(L.wave);
    for (int base = dvs_bid() * NW; base < B; base += gridDim.x * NW) {
        const int dag = base + L.wave;
        const bool live = dag < B;
        const size_t dg = live ? dag : 0;
        const int Nl = live ? N : 0;
        const uint32_t gdag = a.dims.dag_offset + (uint32_t)dg;
        const uint32_t kprob = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_prob, gdag);
        const unsigned allowed_r = live ? a.rec[dg].allowed[L.r] : (1u << L.r);
        f4 q[4], k[4], v[4];
        {
            f4 x[4], kv[4], dummy[4];
            float rstd;
            {
                dvs_load_x<false>(x, dummy, rstd, a.xin, a.ln, l.lg, l.lb, dg, Nl, L);
                if (a.kv) dvs_load_tile(kv, a.kv, dg, L);
            }
            if (!a.kv) {
#pragma unroll
                for (int t = 0; t < 4; ++t) kv[t] = x[t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                q[t] = dvs_vecT(l.inb, t, L);
                k[t] = dvs_vecT(l.inb + 64, t, L);
                v[t] = f4_splat(l.inb[128 + 16 * t + L.r]);
            }
            {
                const SplitT xs = dvs_split_T(x);
                dvs_matb_T<4>(q, xs, l.Winh, l.Winl, 0, L);
                if (a.kv) {
                    const SplitT ks = dvs_split_T(kv);
                    dvs_matb_T<4>(k, ks, l.Winh, l.Winl, 64, L);
                    dvs_matb_N<4>(v, ks, l.Winh, l.Winl, 128, L);
                } else {
                    dvs_matb_T<4>(k, xs, l.Winh, l.Winl, 64, L);
                    dvs_matb_N<4>(v, xs, l.Winh, l.Winl, 128, L);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] *= scale;
        }
        f4 qN[4], kN[4], vT[4], dOT[4], dON[4];
        {
            f4 dy[4];
            dvs_load_grad(dy, a.gpre, dg, Nl, L);
            dvs_dropout_tile(dy, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L);
            dvs_park_bf((dvs_bf16*)sA, dy, L);         // [hi | lo] bf16, stays parked until the cooperative dWo below
#pragma unroll
            for (int t = 0; t < 4; ++t) dOT[t] = f4_zero();
            dvs_matb_T<4>(dOT, dvs_split_T(dy), l.WoTh, l.WoTl, 0, L);
        }
        dvs_t2n<4>(qN, q, sB, L);
        dvs_t2n<4>(kN, k, sB, L);
        dvs_n2t<4>(vT, v, sB, L);
        dvs_t2n<4>(dON, dOT, sB, L);

        const ProbMask pm = dvs_prob_mask(kprob, D, L);
        const bool hsel = ((L.r & 3) >> 1) != 0;     // N-layout lane r holds slot r = feature 4(r&3) + (r>>2): head bit
        bool ok[4];
        unsigned al4[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            ok[reg] = (allowed_r >> (4 * L.g + reg)) & 1u;
            al4[reg] = (unsigned)__shfl((int)allowed_r, 4 * L.g + reg);
        }
        // One feature tile (= one head pair) per pass: the smallest set of live temporaries; the second wave on the SIMD
        // supplies the instruction-level parallelism that a wider pass would.
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            // ---- T orientation: reg <-> (query i = r, key j = 4g+reg) --------------------------------------------
            f4 pT[2], dsT[2], ds[2], pd[2];
            float delta[2];
            {
                f4 sT[2] = {f4_zero(), f4_zero()};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    sT[0] = dvs_mfma(k[t][kk], q[t][kk], sT[0]);
                    sT[1] = dvs_mfma(k[t][kk + 2], q[t][kk + 2], sT[1]);
                }
                float m[2], den[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) mx = ok[reg] ? fmaxf(mx, sT[u][reg]) : mx;
                    m[u] = mx;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = dvs_max_x16(m[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = dvs_max_x32(m[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float sum = 0.f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        pT[u][reg] = ok[reg] ? __expf(sT[u][reg] - m[u]) : 0.f;
                        sum += pT[u][reg];
                    }
                    den[u] = sum;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) den[u] = dvs_add_x16(den[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) den[u] = dvs_add_x32(den[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) pT[u] *= dvs_rcp(den[u]);      // (1 ulp: a gradient term, dvs_device.h)
            }
            {
                f4 mk[2], dpT[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    mk[u] = mask_T(pm, 2 * t + u, D);
                    dpT[u] = f4_zero();
                }
                // O = P' V (N-layout, columns = slots, per-lane head select), parked row-major in slot B for dWo
                {
                    f4 oa = f4_zero(), ob = f4_zero();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        oa = dvs_mfma(pT[0][kk] * mk[0][kk], v[t][kk], oa);
                        ob = dvs_mfma(pT[1][kk] * mk[1][kk], v[t][kk], ob);
                    }
                    dvs_bf16* po = (dvs_bf16*)sB + (4 * L.g) * DVS_PLD + 16 * t + L.r;       // [hi | lo] bf16 image of O
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        dvs_bf16 oh, ol;
                        dvs_split1(hsel ? ob[reg] : oa[reg], oh, ol);
                        po[reg * DVS_PLD] = oh;
                        po[DVS_PKB + reg * DVS_PLD] = ol;
                    }
                }
                // dP^T = V dO^T
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    dpT[0] = dvs_mfma(vT[t][kk], dOT[t][kk], dpT[0]);
                    dpT[1] = dvs_mfma(vT[t][kk + 2], dOT[t][kk + 2], dpT[1]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dpT[u] *= mk[u];
                    float dl = 0.f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) dl += pT[u][reg] * dpT[u][reg];
                    delta[u] = dl;
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) delta[u] = dvs_add_x16(delta[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u) delta[u] = dvs_add_x32(delta[u]);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) dsT[u][reg] = pT[u][reg] * (dpT[u][reg] - delta[u]);
                // dk and dv contract over the QUERIES: they need dS and P' with lane r <-> key j, register <-> query 4g + reg.
                // The four 16 x 16 tiles of the head pair go through the wave's transposing tile, one after the other (the LDS
                // queue of a wave is in order: the next tile's write is queued behind this one's reads; one wait for all four) —
                // rounds 1-3 recomputed scores, exponentials and masks in that orientation (8 MFMAs and ~100 vector
                // instructions per pass, the row statistics exchanged through LDS)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    *(f4*)(st + L.r * DVS_ATTNB_TLD + 4 * L.g) = dsT[u];
                    dvs_wave_sync();
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) ds[u][reg] = st[(4 * L.g + reg) * DVS_ATTNB_TLD + L.r];
                    dvs_wave_sync();
                    *(f4*)(st + L.r * DVS_ATTNB_TLD + 4 * L.g) = pT[u] * mk[u];
                    dvs_wave_sync();
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) pd[u][reg] = st[(4 * L.g + reg) * DVS_ATTNB_TLD + L.r];
                    dvs_wave_sync();
                }
            }
            // dq^T = K^T dS^T: all 16 slot rows per head, merged by register; stored at once (scaled by 1/sqrt(dh))
            {
                f4 qa = f4_zero(), qb = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    qa = dvs_mfma(kN[t][kk], dsT[0][kk], qa);
                    qb = dvs_mfma(kN[t][kk], dsT[1][kk], qb);
                }
                const f4 dq = f4{qa[0], qa[1], qb[2], qb[3]} * scale;
                if (live) ((f4*)(a.gq + dg * DVS_TILE))[t * 64 + L.lane] = dq;
            }
            // ---- S orientation (reg <-> query i = 4g+reg, lane r <-> key j): ds, pd from the transposes above ------------
            {
                // dk^T = Q^T dS ;  dv^T = dO^T P'  (per head on all slot rows, merged by register), stored at once
                f4 ka = f4_zero(), kb = f4_zero(), va = f4_zero(), vb = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    ka = dvs_mfma(qN[t][kk], ds[0][kk], ka);
                    kb = dvs_mfma(qN[t][kk], ds[1][kk], kb);
                    va = dvs_mfma(dON[t][kk], pd[0][kk], va);
                    vb = dvs_mfma(dON[t][kk], pd[1][kk], vb);
                }
                if (live) {
                    ((f4*)(a.gk + dg * DVS_TILE))[t * 64 + L.lane] = f4{ka[0], ka[1], kb[2], kb[3]};
                    ((f4*)(a.gv + dg * DVS_TILE))[t * 64 + L.lane] = f4{va[0], va[1], vb[2], vb[3]};
                }
            }
            DVS_SCHED_FENCE();
        }
        // ---- dWo += dy^T O over the group's DAGs ------------------------------------------------------------------
        dvs_group_barrier(G, L);
        dvs_coop_dw_bf(aWo, abo, (const dvs_bf16*)l.slots, (const dvs_bf16*)l.slots + 2 * DVS_SCR, 2 * 2 * DVS_SCR, L);
        dvs_group_barrier(G, L);
        if (base == dvs_bid() * NW) DVS_STAMP(dvs_stamps_bwd, mine, 7);
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 2);
    // every wave touches its first tiles of the next phase (dvs_stage.h) — BEHIND the older group's image prefetch: ahead of it
    // the cold reads delay the issue of the 73 KB of image loads, which for the short phases sits on the critical path
    // (DVS_TOUCH_MODE: 0 none, 1 ahead of the prefetch, 2 behind it; A/B builds)
#if DVS_TOUCH_MODE == 1
    const DvsTouch touch = dvs_touch_first(next, has_next, NW, L.wave, L.lane);
#endif
    DvsBwdTail tail;
    dvs_tail_issue(tail, next, has_next);
#if DVS_TOUCH_MODE == 2
    const DvsTouch touch = dvs_touch_first(next, has_next, NW, L.wave, L.lane);
#elif DVS_TOUCH_MODE == 0
    const DvsTouch touch = {{0.f, 0.f}};
#endif
    DVS_STAMP(dvs_stamps_bwd, mine, 3);
    dvs_lds_barrier();
    DVS_STAMP(dvs_stamps_bwd, mine, 4);
    float* slab = a.slab + (size_t)dvs_bid() * a.P;
    float* const bufo = dvs_bwd_epi(smem);
    float* red = bufo + 4096;
    dvs_coop_stage<NW>(bufo, aWo, L);
    red[L.wave * 64 + L.lane] = 0.f;
    dvs_wave_sync();
    if (L.r == 0) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) red[L.wave * 64 + 16 * (L.wave & 3) + 4 * L.g + reg] = abo[reg];
    }
    dvs_lds_barrier();
    dvs_coop_flush<NW>(bufo, slab + a.o_out_w, aWo, L, false, true);     // columns back to parameter order
    if (dvs_tid() < 64) {
        float s = 0.f;
        for (int w = 0; w < NW; ++w) s += red[w * 64 + dvs_tid()];
        slab[a.o_out_b + dvs_tid()] = s;
    }
    DVS_STAMP(dvs_stamps_bwd, mine, 5);
    dvs_tail_commit(tail, next, has_next, smem);
    DVS_STAMP(dvs_stamps_bwd, mine, 6);
    dvs_touch_done(touch);
}

