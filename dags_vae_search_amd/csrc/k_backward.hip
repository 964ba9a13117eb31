// Backward kernels: FFN sublayer, stacked-projection (q/k/v) backward with fused LayerNorm backward, slab reduce.
#include "dvs_backward.h"

// ---------------------------------------------------------------------------------------------------------
// FFN sublayer backward (autograd of pace.py:62-65 / 151-153).  Recomputes h = drop(relu(W1 x + b1)) from the saved
// pre-sum of the producing sublayer; all four products (dW2, dh, dW1, dx) are MFMA chains on registers.
// ---------------------------------------------------------------------------------------------------------
struct FfnBLds {
    float *W1, *W2, *b1, *b2, *lg, *lb, *og, *ob, *scr;
};
__device__ __forceinline__ FfnBLds ffnb_lds(char* smem) {
    FfnBLds l;
    l.W1 = (float*)smem;
    l.W2 = l.W1 + 64 * DVS_LD;
    l.b1 = l.W2 + 64 * DVS_LD;
    l.b2 = l.b1 + 64;
    l.lg = l.b2 + 64;
    l.lb = l.lg + 64;
    l.og = l.lb + 64;
    l.ob = l.og + 64;
    l.scr = l.ob + 64;
    return l;
}
static size_t ffnb_lds_floats(int nwaves) { return 128 * DVS_LD + 6 * 64 + (size_t)nwaves * (DVS_SCR + 3 * DVS_TILE); }

__global__ __launch_bounds__(256) void k_ffn_bwd(FfnBwdArgs a) {
    DVS_DYN_LDS(smem);
    const FfnBLds l = ffnb_lds(smem);
    dvs_stage_matrix(l.W1, DVS_LD, a.l1_w, 64, 64, 64);
    dvs_stage_matrix(l.W2, DVS_LD, a.l2_w, 64, 64, 64);
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
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N;
    float* scr = l.scr + L.wave * DVS_SCR;
    float* pf = l.scr + L.nwaves * DVS_SCR + L.wave * 3 * DVS_TILE;    // LDS-DMA landing zone: x, d pre, own pre
    f4 dW1[4][4], dW2[4][4], db1[4], db2[4], dgam[4], dbet[4], dog[4], dob[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        db1[i] = db2[i] = dgam[i] = dbet[i] = dog[i] = dob[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) dW1[i][j] = dW2[i][j] = f4_zero();
    }
    const int Bl = a.dims.B;
    const int stride = gridDim.x * L.nwaves;
    int dag = blockIdx.x * L.nwaves + L.wave;
    if (dag < Bl) {
        dvs_prefetch_tile(pf, a.xin, dag, L);
        dvs_prefetch_tile(pf + DVS_TILE, a.gpre, dag, L);
        if (a.own_pre) dvs_prefetch_tile(pf + 2 * DVS_TILE, a.own_pre, dag, L);
    }
    for (; dag < Bl; dag += stride) {
        dvs_prefetch_wait();
        f4 x[4], xhat[4];
        float rstd;
        dvs_load_x<true>(x, xhat, rstd, a.xin, a.ln, l.lg, l.lb, dag, N, L, pf);
        f4 gp[4];
        dvs_load_grad(gp, a.gpre, dag, N, L, pf + DVS_TILE);
        f4 po[4], pxh[4];
        float prstd = 1.f;
        if (a.own_pre) dvs_load_x<true>(po, pxh, prstd, a.own_pre, a.own, l.og, l.ob, dag, N, L, pf + 2 * DVS_TILE);
        dvs_slot_release();
        if (dag + stride < Bl) {
            dvs_prefetch_tile(pf, a.xin, dag + stride, L);
            dvs_prefetch_tile(pf + DVS_TILE, a.gpre, dag + stride, L);
            if (a.own_pre) dvs_prefetch_tile(pf + 2 * DVS_TILE, a.own_pre, dag + stride, L);
        }
        if (a.own_pre) dvs_ln_bwd(gp, pxh, prstd, l.og, dog, dob, L);   // d(LN_own(pre_own)) -> d(pre_own)
        const uint32_t gdag = a.dims.dag_offset + dag;
        const uint32_t khid = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_hidden, gdag);
        const uint32_t kpost = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag);
        // recompute hidden
        f4 hpre[4], hd[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) hpre[t] = dvs_vecT(l.b1, t, L);
        dvs_mat_T<4, 4>(hpre, x, l.W1, DVS_LD, 0, L);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) hd[t][kk] = fmaxf(hpre[t][kk], 0.f);
        dvs_dropout_tile(hd, khid, D, L);
        // dy = d(W2 h + b2) = dropout-mask(post) applied to d pre
        f4 dy[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dy[t] = gp[t];
        dvs_dropout_tile(dy, kpost, D, L);
#pragma unroll
        for (int t = 0; t < 4; ++t) db2[t] += dy[t];
        f4 dyN[4], hN[4];
        dvs_t2n<4>(dyN, dy, scr, L);
        dvs_t2n<4>(hN, hd, scr, L);
        dvs_outer_acc<4, 4>(dW2, dyN, hN);
        f4 dh[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
        dvs_mat_Tt<4, 4>(dh, dy, l.W2, DVS_LD, 0, L);
        dvs_dropout_tile(dh, khid, D, L);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dh[t][kk] = hpre[t][kk] > 0.f ? dh[t][kk] : 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) db1[t] += dh[t];
        f4 dhN[4], xN[4];
        dvs_t2n<4>(dhN, dh, scr, L);
        dvs_t2n<4>(xN, x, scr, L);
        dvs_outer_acc<4, 4>(dW1, dhN, xN);
        f4 dx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) dx[t] = gp[t];
        dvs_mat_Tt<4, 4>(dx, dh, l.W1, DVS_LD, 0, L);
        if (a.ln.stats) dvs_ln_bwd(dx, xhat, rstd, l.lg, dgam, dbet, L);
        dvs_store_tile(a.gout, dag, dx, L);
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    float* rW1 = (float*)smem;
    float* rW2 = rW1 + DVS_RED_MAT;
    float* rv = rW2 + DVS_RED_MAT;               // 6 vectors
    float* es = rv + 6 * DVS_RED_VEC + L.wave * DVS_SCR;
    dvs_stage_dw<4, 4>(rW1, dW1, L);
    dvs_stage_dw<4, 4>(rW2, dW2, L);
    dvs_stage_vec<4>(rv, db1, es, L);
    dvs_stage_vec<4>(rv + DVS_RED_VEC, db2, es, L);
    dvs_stage_vec<4>(rv + 2 * DVS_RED_VEC, dgam, es, L);
    dvs_stage_vec<4>(rv + 3 * DVS_RED_VEC, dbet, es, L);
    dvs_stage_vec<4>(rv + 4 * DVS_RED_VEC, dog, es, L);
    dvs_stage_vec<4>(rv + 5 * DVS_RED_VEC, dob, es, L);
    __syncthreads();
    dvs_flush_dw<4, 4>(rW1, slab + a.o_l1_w, L);
    dvs_flush_dw<4, 4>(rW2, slab + a.o_l2_w, L);
    dvs_flush_vec<4>(rv, slab + a.o_l1_b, L);
    dvs_flush_vec<4>(rv + DVS_RED_VEC, slab + a.o_l2_b, L);
    if (a.o_ln_g >= 0) {
        dvs_flush_vec<4>(rv + 2 * DVS_RED_VEC, slab + a.o_ln_g, L);
        dvs_flush_vec<4>(rv + 3 * DVS_RED_VEC, slab + a.o_ln_b, L);
    }
    if (a.o_own_g >= 0) {
        dvs_flush_vec<4>(rv + 4 * DVS_RED_VEC, slab + a.o_own_g, L);
        dvs_flush_vec<4>(rv + 5 * DVS_RED_VEC, slab + a.o_own_b, L);
    }
}

void dvs_launch_ffn_bwd(const FfnBwdArgs& a, int grid, dvs_stream_t st) {
    size_t lds = ffnb_lds_floats(4) * 4;
    const size_t red = (2 * DVS_RED_MAT + 6 * DVS_RED_VEC + 4 * DVS_SCR) * 4;
    if (lds < red) lds = red;
    DVS_SET_LDS(k_ffn_bwd, lds);
    DVS_LAUNCH(k_ffn_bwd, dim3(grid), dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Backward of NPROJ stacked 64->64 projections of one input X (the q/k/v in-projections of nn.MultiheadAttention):
//   dX^T = sum_p W_p^T dY_p^T (+ residual) ; dW_p += dY_p(N) (x) X(N) ; db_p += sum_tok dY_p ; then the producing
//   sublayer's LayerNorm backward.  Used for self-attention (NPROJ=3), cross-attention q (1) and k,v (2, X = memory).
// ---------------------------------------------------------------------------------------------------------
template <int NPROJ>
__global__ __launch_bounds__(256) void k_proj_bwd(ProjBwdArgs a) {
    DVS_DYN_LDS(smem);
    float* W = (float*)smem;                       // [64*NPROJ][LD]
    float* lg = W + 64 * NPROJ * DVS_LD;
    float* lb = lg + 64;
    float* scr0 = lb + 64;
    if (a.slot_order) dvs_stage_matrix_perm(W, DVS_LD, a.w, 64, 64 * NPROJ, 64, true, false);
    else dvs_stage_matrix(W, DVS_LD, a.w, 64, 64 * NPROJ, 64);
    if (a.ln.stats) {
        dvs_stage_vector(lg, a.ln.g, 64);
        dvs_stage_vector(lb, a.ln.b, 64);
    }
    __syncthreads();
    const Lane L = dvs_lane();
    const int N = a.dims.N;
    float* scr = scr0 + L.wave * DVS_SCR;
    float* pf = scr0 + L.nwaves * DVS_SCR + L.wave * (NPROJ + 2) * DVS_TILE;   // LDS-DMA landing zone: x, residual, dY_p
    f4 dW[NPROJ][4][4], db[NPROJ][4], dgam[4], dbet[4];
#pragma unroll
    for (int p = 0; p < NPROJ; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            db[p][i] = f4_zero();
#pragma unroll
            for (int j = 0; j < 4; ++j) dW[p][i][j] = f4_zero();
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) dgam[i] = dbet[i] = f4_zero();
    const int stride = gridDim.x * L.nwaves;
    int dag = blockIdx.x * L.nwaves + L.wave;
    auto request = [&](int d) {
        dvs_prefetch_tile(pf, a.xin, d, L);
        if (a.gres) dvs_prefetch_tile(pf + DVS_TILE, a.gres, d, L);
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) dvs_prefetch_tile(pf + (2 + p) * DVS_TILE, a.gy[p], d, L);
    };
    if (dag < a.dims.B) request(dag);
    for (; dag < a.dims.B; dag += stride) {
        dvs_prefetch_wait();
        f4 x[4], xhat[4];
        float rstd;
        dvs_load_x<true>(x, xhat, rstd, a.xin, a.ln, lg, lb, dag, N, L, pf);
        f4 dy[NPROJ][4], dx[4];
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) dvs_load_grad(dy[p], a.gy[p], dag, N, L, pf + (2 + p) * DVS_TILE);
        if (a.gres) {
            dvs_load_grad(dx, a.gres, dag, N, L, pf + DVS_TILE);
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) dx[t] = f4_zero();
        }
        dvs_slot_release();
        if (dag + stride < a.dims.B) request(dag + stride);
        f4 xN[4];
        dvs_t2n<4>(xN, x, scr, L);
#pragma unroll
        for (int p = 0; p < NPROJ; ++p) {
            f4 dyN[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) db[p][t] += dy[p][t];
            dvs_t2n<4>(dyN, dy[p], scr, L);
            dvs_outer_acc<4, 4>(dW[p], dyN, xN);
            dvs_mat_Tt<4, 4>(dx, dy[p], W, DVS_LD, 64 * p, L);
        }
        if (a.ln.stats) dvs_ln_bwd(dx, xhat, rstd, lg, dgam, dbet, L);
        if (a.accumulate_out) {
            f4 old[4];
            dvs_load_tile(old, a.gout, dag, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) dx[t] += old[t];
        }
        dvs_store_tile(a.gout, dag, dx, L);
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    float* rW = (float*)smem;                     // up to 2 matrices per pass
    float* rv = rW + 2 * DVS_RED_MAT;             // NPROJ + 2 vectors
    float* es = rv + 5 * DVS_RED_VEC + L.wave * DVS_SCR;
#pragma unroll
    for (int p = 0; p < NPROJ; ++p) dvs_stage_vec<4>(rv + p * DVS_RED_VEC, db[p], es, L);
    dvs_stage_vec<4>(rv + NPROJ * DVS_RED_VEC, dgam, es, L);
    dvs_stage_vec<4>(rv + (NPROJ + 1) * DVS_RED_VEC, dbet, es, L);
#pragma unroll
    for (int p0 = 0; p0 < NPROJ; p0 += 2) {
        if (p0 > 0) __syncthreads();
        dvs_stage_dw<4, 4>(rW, dW[p0], L);
        if (p0 + 1 < NPROJ) dvs_stage_dw<4, 4>(rW + DVS_RED_MAT, dW[p0 + 1 < NPROJ ? p0 + 1 : p0], L);
        __syncthreads();
        const bool so = a.slot_order != 0;
        dvs_flush_dw<4, 4>(rW, slab + a.o_w + 4096 * p0, L, 64, 64, 64, so, false);
        if (p0 + 1 < NPROJ) dvs_flush_dw<4, 4>(rW + DVS_RED_MAT, slab + a.o_w + 4096 * (p0 + 1), L, 64, 64, 64, so, false);
        if (p0 == 0) {
#pragma unroll
            for (int p = 0; p < NPROJ; ++p) dvs_flush_vec<4>(rv + p * DVS_RED_VEC, slab + a.o_b + 64 * p, L, 64, so);
            if (a.o_ln_g >= 0) {
                dvs_flush_vec<4>(rv + NPROJ * DVS_RED_VEC, slab + a.o_ln_g, L);
                dvs_flush_vec<4>(rv + (NPROJ + 1) * DVS_RED_VEC, slab + a.o_ln_b, L);
            }
        }
    }
}

void dvs_launch_proj_bwd(const ProjBwdArgs& a, int nproj, int grid, dvs_stream_t st) {
    const size_t lds = ((size_t)64 * nproj * DVS_LD + 128 + 4 * DVS_SCR + 4 * (size_t)(nproj + 2) * DVS_TILE) * 4;
    const size_t lds_min = (2 * DVS_RED_MAT + 5 * DVS_RED_VEC + 4 * DVS_SCR) * 4;   // epilogue staging
    const size_t bytes = lds > lds_min ? lds : lds_min;
    if (nproj == 3) {
        DVS_SET_LDS(k_proj_bwd<3>, bytes);
        DVS_LAUNCH(k_proj_bwd<3>, dim3(grid), dim3(256), bytes, st, a);
    } else if (nproj == 2) {
        DVS_SET_LDS(k_proj_bwd<2>, bytes);
        DVS_LAUNCH(k_proj_bwd<2>, dim3(grid), dim3(256), bytes, st, a);
    } else {
        DVS_SET_LDS(k_proj_bwd<1>, bytes);
        DVS_LAUNCH(k_proj_bwd<1>, dim3(grid), dim3(256), bytes, st, a);
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
