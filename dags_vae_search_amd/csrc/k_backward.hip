// Backward kernels: FFN sublayer, stacked-projection (q/k/v) backward with fused LayerNorm backward, slab reduce.
#include "dvs_backward.h"
#include "dvs_wimg.h"

// ---------------------------------------------------------------------------------------------------------
// FFN sublayer backward (autograd of pace.py:62-65 / 151-153).  Recomputes h = drop(relu(W1 x + b1)) from the saved
// pre-sum of the producing sublayer; all four products (dW2, dh, dW1, dx) are MFMA chains on registers.
// ---------------------------------------------------------------------------------------------------------
struct FfnBLds {
    // bf16x3 images (dvs_bf16.h) of W2^T and W1^T (d hidden, d x); W1 as the bf16x6 triple k_ffn_fwd uses: the hidden is
    // recomputed with the forward's own instruction sequence, because its sign must reproduce the forward's ReLU mask.
    // The weight GRADIENTS (dvs_coop_dw) stay exact fp32.
    dvs_bf16 *W2Th, *W2Tl, *W1Th, *W1Tl, *W1x6;
    float *b1, *b2, *lg, *lb, *og, *ob, *slots;
};
__device__ __forceinline__ FfnBLds ffnb_lds(char* smem) {
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
    l.slots = l.ob + 64;
    return l;
}
static size_t ffnb_lds_bytes() {
    return 7 * 64 * DVS_LDB * sizeof(dvs_bf16) + (6 * 64 + (size_t)8 * 2 * DVS_SCR + 16) * sizeof(float);
}

// 8 waves per workgroup, one DAG per wave per iteration; weight gradients are accumulated cooperatively
// (dvs_coop_dw): ~150 registers per lane, two waves per SIMD, so one wave's VALU phases overlap the other's MFMAs.
// The two gradient products (d hidden, d x) run on the bf16 matrix pipe as bf16x3: gradient
// parity is bounded at 2e-3 of the tensor maximum (tests), three orders of magnitude above their ~1e-5 error, whereas
// the forward keeps exact fp32 MFMAs for the 1e-4 ELBO contract.
__global__ __launch_bounds__(512) void k_ffn_bwd(FfnBwdArgs a) {
    DVS_DYN_LDS(smem);
    const FfnBLds l = ffnb_lds(smem);
    dvs_copy_image(l.W2Th, (const dvs_bf16*)a.wimg + DvsFfnImg::W2T, (int)(4 * DVS_IMG64));   // W2^T, W1^T x3 pairs
    dvs_copy_image(l.W1x6, (const dvs_bf16*)a.wimg + DvsFfnImg::W1, (int)(3 * DVS_IMG64));
    dvs_stage_vector(l.b1, a.l1_b, 64);
    dvs_stage_vector(l.b2, a.l2_b, 64);
    if (a.ln.stats) {
        dvs_stage_vector(l.lg, a.ln.g, 64);
        dvs_stage_vector(l.lb, a.ln.b, 64);
    }
    if (a.own_pre) {
        dvs_stage_vector(l.og, a.own.g, 64);
        dvs_stage_vector(l.ob, a.own.b, 64);
    }
    int* gcount = (int*)(l.slots + 8 * 2 * DVS_SCR);
    if (threadIdx.x < 2) gcount[threadIdx.x] = 0;
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int B = a.dims.B * a.dims.NT;                    // tiles (dvs_tile_of): the sublayer is token-local
    float* sA = l.slots + L.wave * 2 * DVS_SCR;
    float* sB = sA + DVS_SCR;
    DvsGroup G = {gcount + (L.wave >> 2), 0};
    f4 aW1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()}, aW2[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
    float vb1 = 0.f, vb2 = 0.f, vgam = 0.f, vbet = 0.f, vog = 0.f, vob = 0.f;    // lane = feature
    for (int base = blockIdx.x * 8; base < B; base += gridDim.x * 8) {
        const int dag = base + L.wave;                     // tile index
        const bool live = dag < B;
        const size_t dg = live ? dag : 0;
        const DvsTile T = dvs_tile_of((int)dg, a.dims);
        const int N = T.Nl;
        const int Nl = live ? N : 0;                       // a wave without a tile carries all-zero tiles
        f4 x[4], xhat[4], gp[4];
        float rstd;
        dvs_load_x<true>(x, xhat, rstd, a.xin, a.ln, l.lg, l.lb, dg, Nl, L);
        dvs_load_grad(gp, a.gpre, dg, Nl, L);
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
        dvs_dropout_tile(hd, khid, D, L, T.tok0);
        // dy = d(W2 h + b2) = dropout-mask(post) applied to d pre
        f4 dy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dy[t] = gp[t];
        dvs_dropout_tile(dy, kpost, D, L, T.tok0);
        // ---- dW2 += dy^T hd, db2 += sum dy --------------------------------------------------------------------------
        dvs_park_T(sA, dy, L);
        dvs_park_T(sB, hd, L);
        dvs_wave_sync();
        vb2 += dvs_colsum(sA, L);
        dvs_group_barrier(G, L);
        dvs_coop_dw(aW2, l.slots, l.slots + DVS_SCR, 2 * DVS_SCR, L);
        f4 dh[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
        dvs_matb_T<4>(dh, dvs_split_T(dy), l.W2Th, l.W2Tl, 0, L);
        dvs_dropout_tile(dh, khid, D, L, T.tok0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dh[t][kk] = hpre[t][kk] > 0.f ? dh[t][kk] : 0.f;
        dvs_group_barrier(G, L);
        // ---- dW1 += dh^T x, db1 += sum dh ----------------------------------------------------------------------------
        dvs_park_T(sA, dh, L);
        dvs_park_T(sB, x, L);
        dvs_wave_sync();
        vb1 += dvs_colsum(sA, L);
        dvs_group_barrier(G, L);
        dvs_coop_dw(aW1, l.slots, l.slots + DVS_SCR, 2 * DVS_SCR, L);
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
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    dvs_coop_store((float*)smem, slab + a.o_l1_w, aW1, L);
    dvs_coop_store((float*)smem, slab + a.o_l2_w, aW2, L);
    float* red = (float*)smem;                        // [8 waves][6][64]
    red[(L.wave * 6 + 0) * 64 + L.lane] = vb1;
    red[(L.wave * 6 + 1) * 64 + L.lane] = vb2;
    red[(L.wave * 6 + 2) * 64 + L.lane] = vgam;
    red[(L.wave * 6 + 3) * 64 + L.lane] = vbet;
    red[(L.wave * 6 + 4) * 64 + L.lane] = vog;
    red[(L.wave * 6 + 5) * 64 + L.lane] = vob;
    __syncthreads();
    if (threadIdx.x < 6 * 64) {
        const int k = threadIdx.x >> 6, f = threadIdx.x & 63;
        float s = 0.f;
        for (int w = 0; w < 8; ++w) s += red[(w * 6 + k) * 64 + f];
        const int64_t off = k == 0 ? a.o_l1_b : k == 1 ? a.o_l2_b : k == 2 ? a.o_ln_g : k == 3 ? a.o_ln_b : k == 4 ? a.o_own_g : a.o_own_b;
        if (off >= 0) slab[off + f] = s;
    }
}

void dvs_launch_ffn_bwd(const FfnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = ffnb_lds_bytes();
    DVS_SET_LDS(k_ffn_bwd, lds);
    DVS_LAUNCH(k_ffn_bwd, dim3(grid), dim3(512), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Backward of NPROJ stacked 64->64 projections of one input X (the q/k/v in-projections of nn.MultiheadAttention):
//   dX^T = sum_p W_p^T dY_p^T (+ residual) ; dW_p += dY_p(N) (x) X(N) ; db_p += sum_tok dY_p ; then the producing
//   sublayer's LayerNorm backward.  Used for self-attention (NPROJ=3), cross-attention q (1) and k,v (2, X = memory).
// ---------------------------------------------------------------------------------------------------------
// 8 waves per workgroup in two independent groups of four; weight gradients accumulated cooperatively (dvs_coop_dw):
// per wave 16 accumulator registers per projection, ~130 VGPRs, two waves per SIMD.  LDS slots per wave: X (kept for
// all projections of the DAG) and two alternating dY slots, so one group barrier per projection + one per DAG.
template <int NPROJ>
__global__ __launch_bounds__(512) void k_proj_bwd(ProjBwdArgs a) {
    DVS_DYN_LDS(smem);
    // W_p^T as bf16x3 images (dvs_bf16.h): dX^T = sum_p W_p^T dY_p^T is a pure gradient product (no mask or statistic
    // of the forward depends on it), so it runs on the bf16 matrix pipe; the weight gradients stay exact fp32.
    dvs_bf16* WT = (dvs_bf16*)smem;                // [NPROJ][hi | lo][64][LDB]
    float* lg = (float*)(WT + NPROJ * 2 * DVS_IMG64);
    float* lb = lg + 64;
    float* slots = lb + 64;                        // per wave 3 tiles: A0, A1 (alternating dY) and B (X)
    int* gcount = (int*)(slots + 8 * 3 * DVS_SCR);
    dvs_copy_image(WT, (const dvs_bf16*)a.wimg, (int)(NPROJ * 2 * DVS_IMG64));
    if (a.ln.stats) {
        dvs_stage_vector(lg, a.ln.g, 64);
        dvs_stage_vector(lb, a.ln.b, 64);
    }
    if (threadIdx.x < 2) gcount[threadIdx.x] = 0;
    __syncthreads();
    const Lane L = dvs_lane();
    const int B = a.dims.B * a.dims.NT;              // tiles
    float* myA0 = slots + L.wave * 3 * DVS_SCR;
    float* myA1 = myA0 + DVS_SCR;
    float* myB = myA0 + 2 * DVS_SCR;
    DvsGroup G = {gcount + (L.wave >> 2), 0};
    f4 aW[NPROJ][4];
    float vb[NPROJ], vgam = 0.f, vbet = 0.f;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) {
        vb[p] = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) aW[p][i] = f4_zero();
    }
    for (int base = blockIdx.x * 8; base < B; base += gridDim.x * 8) {
        const int dag = base + L.wave;               // tile index
        const bool live = dag < B;
        const size_t dg = live ? dag : 0;
        const int Nl = live ? dvs_tile_of((int)dg, a.dims).Nl : 0;
        f4 x[4], xhat[4], dx[4];
        float rstd;
        dvs_load_x<true>(x, xhat, rstd, a.xin, a.ln, lg, lb, dg, Nl, L);
        if (a.gres) {
            dvs_load_grad(dx, a.gres, dg, Nl, L);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) dx[t] = f4_zero();
        }
        dvs_park_T(myB, x, L);
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) {
            float* mine = (p & 1) ? myA1 : myA0;
            f4 dy[4];
            dvs_load_grad(dy, a.gy[p], dg, Nl, L);
            dvs_park_T(mine, dy, L);
            dvs_wave_sync();
            vb[p] += dvs_colsum(mine, L);
            dvs_group_barrier(G, L);
            dvs_coop_dw(aW[p], slots + (p & 1) * DVS_SCR, slots + 2 * DVS_SCR, 3 * DVS_SCR, L);
            dvs_matb_T<4>(dx, dvs_split_T(dy), WT + p * 2 * DVS_IMG64, WT + p * 2 * DVS_IMG64 + DVS_IMG64, 0, L);
        }
        dvs_group_barrier(G, L);        // every wave of the group is done with this DAG's slots
        if (a.ln.stats) {
            f4 t0[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) t0[t] = dx[t] * xhat[t];
            dvs_park_T(myA0, t0, L);
            dvs_park_T(myB, dx, L);
            dvs_wave_sync();
            vgam += dvs_colsum(myA0, L);
            vbet += dvs_colsum(myB, L);
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
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    const bool so = a.slot_order != 0;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) dvs_coop_store((float*)smem, slab + a.o_w + 4096 * p, aW[p], L, so, false);
    float* red = (float*)smem;                        // [8 waves][NPROJ + 2][64]
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) red[(L.wave * (NPROJ + 2) + p) * 64 + L.lane] = vb[p];
    red[(L.wave * (NPROJ + 2) + NPROJ) * 64 + L.lane] = vgam;
    red[(L.wave * (NPROJ + 2) + NPROJ + 1) * 64 + L.lane] = vbet;
    __syncthreads();
    if (threadIdx.x < (NPROJ + 2) * 64) {
        const int k = threadIdx.x >> 6, f = threadIdx.x & 63;
        float s = 0.f;
        for (int w = 0; w < 8; ++w) s += red[(w * (NPROJ + 2) + k) * 64 + f];
        if (k < NPROJ) slab[a.o_b + 64 * k + (so ? dvs_pi(f) : f)] = s;
        else if (a.o_ln_g >= 0) slab[(k == NPROJ ? a.o_ln_g : a.o_ln_b) + f] = s;
    }
}

void dvs_launch_proj_bwd(const ProjBwdArgs& a, int nproj, int grid, dvs_stream_t st) {
    const size_t bytes = (size_t)2 * nproj * 64 * DVS_LDB * sizeof(dvs_bf16) + (128 + (size_t)8 * 3 * DVS_SCR + 16) * 4;
    if (nproj == 3) {
        DVS_SET_LDS(k_proj_bwd<3>, bytes);
        DVS_LAUNCH(k_proj_bwd<3>, dim3(grid), dim3(512), bytes, st, a);
    } else if (nproj == 2) {
        DVS_SET_LDS(k_proj_bwd<2>, bytes);
        DVS_LAUNCH(k_proj_bwd<2>, dim3(grid), dim3(512), bytes, st, a);
    } else {
        DVS_SET_LDS(k_proj_bwd<1>, bytes);
        DVS_LAUNCH(k_proj_bwd<1>, dim3(grid), dim3(512), bytes, st, a);
    }
}

// ---------------------------------------------------------------------------------------------------------
// grads[p] = sum over slabs (fixed order)
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reduce_slabs(ReduceArgs a) {
    const int64_t i4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= a.P) return;
    const bool fc = (i4 >= a.fc_lo1 && i4 < a.fc_hi1) || (i4 >= a.fc_lo2 && i4 < a.fc_hi2);
    const float* src = fc ? a.fcpart : a.slab;
    const int n = fc ? DVS_FC_PARTS : a.nslab;
    f4 s = f4_zero();
    int k = 0;
    for (; k + 8 <= n; k += 8) {      // 8 independent 16-byte loads in flight per lane
        f4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *(const f4*)(src + (size_t)(k + u) * a.P + i4);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < n; ++k) s += *(const f4*)(src + (size_t)k * a.P + i4);
    *(f4*)(a.grads + i4) = s;
}

void dvs_launch_reduce_slabs(const ReduceArgs& a, dvs_stream_t st) {
    const int64_t n4 = (a.P + 3) / 4;
    DVS_LAUNCH(k_reduce_slabs, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, a);
}
