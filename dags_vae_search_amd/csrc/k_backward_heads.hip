// Backward of the loss head, the embedding, and the latent block (fc1/fc2/fc3).
#include "dvs_backward.h"

// ---------------------------------------------------------------------------------------------------------
// Loss head backward (autograd of pace.py:1880-1972) fused with the last decoder LayerNorm's backward.
// ---------------------------------------------------------------------------------------------------------
constexpr int LOSS_LDN2 = 36;
struct LossBLds {
    float *Wn1, *Wn2, *Wa, *Wb, *bn1, *bn2, *be1, *w2, *b2, *lg, *lb, *scr;
};
__device__ __forceinline__ LossBLds lossb_lds(char* smem) {
    LossBLds l;
    l.Wn1 = (float*)smem;
    l.Wn2 = l.Wn1 + 32 * DVS_LD;
    l.Wa = l.Wn2 + 16 * LOSS_LDN2;
    l.Wb = l.Wa + 64 * DVS_LD;
    l.bn1 = l.Wb + 64 * DVS_LD;
    l.bn2 = l.bn1 + 32;
    l.be1 = l.bn2 + 16;
    l.w2 = l.be1 + 64;
    l.b2 = l.w2 + 64;
    l.lg = l.b2 + 16;
    l.lb = l.lg + 64;
    l.scr = l.lb + 64;
    return l;
}

__global__ __launch_bounds__(256) void k_loss_bwd(LossArgs a) {
    DVS_DYN_LDS(smem);
    const LossBLds l = lossb_lds(smem);
    const int N = a.dims.N, C = a.dims.C;
    dvs_stage_matrix(l.Wn1, DVS_LD, a.node0_w, 64, 32, 64);
    for (int i = threadIdx.x; i < 16 * 32; i += blockDim.x) {
        const int c = i >> 5, k = i & 31;
        l.Wn2[c * LOSS_LDN2 + k] = c < C ? a.node2_w[c * 32 + k] : 0.f;
    }
    dvs_stage_matrix(l.Wa, DVS_LD, a.edge0_w, 128, 64, 64);
    dvs_stage_matrix(l.Wb, DVS_LD, a.edge0_w + 64, 128, 64, 64);
    dvs_stage_vector(l.bn1, a.node0_b, 32);
    for (int i = threadIdx.x; i < 16; i += blockDim.x) l.bn2[i] = i < C ? a.node2_b[i] : 0.f;
    dvs_stage_vector(l.be1, a.edge0_b, 64);
    dvs_stage_vector(l.w2, a.edge2_w, 64);
    if (threadIdx.x == 0) l.b2[0] = a.edge2_b[0];
    dvs_stage_vector(l.lg, a.ln.g, 64);
    dvs_stage_vector(l.lb, a.ln.b, 64);
    __syncthreads();
    const Lane L = dvs_lane();
    float* scrV = l.scr + L.wave * 3 * DVS_SCR;
    float* scrU = scrV + DVS_SCR;
    float* dlm = scrU + DVS_SCR;      // [16][16] d logit matrix
    const float b2 = l.b2[0];
    const float gr = a.gcoef[0];
    f4 dWn1[2][4], dWn2[1][2], dWa[4][4], dWb[4][4], dbn1[2], dbn2[1], dbe1[4], dw2[4], dgam[4], dbet[4];
    float db2 = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dbe1[i] = dw2[i] = dgam[i] = dbet[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) dWa[i][j] = dWb[i][j] = f4_zero();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        dbn1[i] = f4_zero();
        dWn2[0][i] = f4_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) dWn1[i][j] = f4_zero();
    }
    dbn2[0] = f4_zero();
    for (int dag = blockIdx.x * L.nwaves + L.wave; dag < a.dims.B; dag += gridDim.x * L.nwaves) {
        f4 h[4], xhat[4];
        float rstd;
        dvs_load_x<true>(h, xhat, rstd, a.xin, a.ln, l.lg, l.lb, dag, N, L);
        const DvsRecord* rec = a.rec + dag;
        f4 hN[4];
        dvs_t2n<4>(hN, h, scrV, L);
        f4 dh[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
        // ---- node head ----------------------------------------------------------------------------------
        {
            f4 t1p[2], t1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) t1p[t] = dvs_vecT(l.bn1, t, L);
            dvs_mat_T<2, 4>(t1p, h, l.Wn1, DVS_LD, 0, L);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) t1[t][kk] = fmaxf(t1p[t][kk], 0.f);
            f4 lgt[1] = {*(const f4*)(l.bn2 + 4 * L.g)};
            dvs_mat_T<1, 2>(lgt, t1, l.Wn2, LOSS_LDN2, 0, L);
            float mx = -3.0e38f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) mx = (4 * L.g + reg < C) ? fmaxf(mx, lgt[0][reg]) : mx;
            mx = dvs_max_g(mx);
            f4 ex;
            float se = 0.f;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                ex[reg] = (4 * L.g + reg < C) ? __expf(lgt[0][reg] - mx) : 0.f;
                se += ex[reg];
            }
            se = dvs_sum_g(se);
            const int target = rec->label[(L.r + 1) & 15];
            const bool vt = L.r < N - 1;
            f4 dlg[1];
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int c = 4 * L.g + reg;
                dlg[0][reg] = (vt && c < C) ? gr * (ex[reg] / se - (c == target ? 1.f : 0.f)) : 0.f;
            }
            dbn2[0] += dlg[0];
            f4 dlgN[1], t1N[2];
            dvs_t2n<1>(dlgN, dlg, scrV, L);
            dvs_t2n<2>(t1N, t1, scrV, L);
            dvs_outer_acc<1, 2>(dWn2, dlgN, t1N);
            f4 dt1[2] = {f4_zero(), f4_zero()};
            dvs_mat_Tt<2, 1>(dt1, dlg, l.Wn2, LOSS_LDN2, 0, L);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dt1[t][kk] = t1p[t][kk] > 0.f ? dt1[t][kk] : 0.f;
                dbn1[t] += dt1[t];
            }
            f4 dt1N[2];
            dvs_t2n<2>(dt1N, dt1, scrV, L);
            dvs_outer_acc<2, 4>(dWn1, dt1N, hN);
            dvs_mat_Tt<4, 2>(dh, dt1, l.Wn1, DVS_LD, 0, L);
        }
        // ---- edge head ----------------------------------------------------------------------------------
        f4 U[4], V[4], w2v[4], dU[4], dV[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            U[t] = f4_zero();
            V[t] = dvs_vecT(l.be1, t, L);
            w2v[t] = dvs_vecT(l.w2, t, L);
            dU[t] = dV[t] = f4_zero();
        }
        dvs_mat_T<4, 4>(U, h, l.Wa, DVS_LD, 0, L);
        dvs_mat_T<4, 4>(V, h, l.Wb, DVS_LD, 0, L);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            *(f4*)(scrV + L.r * DVS_LD + 16 * t + 4 * L.g) = V[t];
            *(f4*)(scrU + L.r * DVS_LD + 16 * t + 4 * L.g) = U[t];
        }
        dvs_wave_sync();
        const unsigned par = rec->parents[(L.r + 1) & 15];
        // pass 1: lane r = i walks j; accumulates dU, dw2, db2; publishes d logit(i, j)
        for (int j = 0; j < N - 2; ++j) {
            f4 pre[4];
            float e = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f4 vj = *(const f4*)(scrV + j * DVS_LD + 16 * t + 4 * L.g);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    pre[t][kk] = fmaxf(U[t][kk] + vj[kk], 0.f);
                    e += w2v[t][kk] * pre[t][kk];
                }
            }
            const float logit = dvs_sum_g(e) + b2;
            const bool pv = (L.r > j) && (L.r <= N - 2);
            const float truth = (float)((par >> (j + 1)) & 1u);
            const float sg = 1.0f / (1.0f + __expf(-logit));
            const float dl = pv ? gr * (sg - truth) : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    dw2[t][kk] += dl * pre[t][kk];
                    dU[t][kk] += pre[t][kk] > 0.f ? dl * w2v[t][kk] : 0.f;
                }
            if (L.g == 0) {
                db2 += dl;
                dlm[L.r * 16 + j] = dl;
            }
        }
        dvs_wave_sync();
        // pass 2: lane r = j walks i; accumulates dV
        for (int i = 1; i <= N - 2; ++i) {
            const float dl = (L.r < i) ? dlm[i * 16 + L.r] : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f4 ui = *(const f4*)(scrU + i * DVS_LD + 16 * t + 4 * L.g);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dV[t][kk] += (ui[kk] + V[t][kk] > 0.f) ? dl * w2v[t][kk] : 0.f;
            }
        }
        dvs_wave_sync();
#pragma unroll
        for (int t = 0; t < 4; ++t) dbe1[t] += dV[t];
        f4 dUN[4], dVN[4];
        dvs_t2n<4>(dUN, dU, scrV, L);
        dvs_t2n<4>(dVN, dV, scrV, L);
        dvs_outer_acc<4, 4>(dWa, dUN, hN);
        dvs_outer_acc<4, 4>(dWb, dVN, hN);
        dvs_mat_Tt<4, 4>(dh, dU, l.Wa, DVS_LD, 0, L);
        dvs_mat_Tt<4, 4>(dh, dV, l.Wb, DVS_LD, 0, L);
        dvs_ln_bwd(dh, xhat, rstd, l.lg, dgam, dbet, L);
        dvs_store_tile(a.gout, dag, dh, L);
    }
    __syncthreads();
    float* buf = (float*)smem;
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    dvs_reduce_dw<2, 4>(buf, dWn1, slab + a.o_node0_w, L);
    dvs_reduce_dw<1, 2>(buf, dWn2, slab + a.o_node2_w, L, C, 32);
    dvs_reduce_dw<4, 4>(buf, dWa, slab + a.o_edge0_w, L, 64, 128);
    dvs_reduce_dw<4, 4>(buf, dWb, slab + a.o_edge0_w + 64, L, 64, 128);
    dvs_reduce_vec<2>(buf, dbn1, slab + a.o_node0_b, L);
    dvs_reduce_vec<1>(buf, dbn2, slab + a.o_node2_b, L, C);
    dvs_reduce_vec<4>(buf, dbe1, slab + a.o_edge0_b, L);
    // dw2 is already a per-(lane c-set) sum over pairs: lanes with different r hold partials of the same feature
    dvs_reduce_vec<4>(buf, dw2, slab + a.o_edge2_w, L);
    dvs_reduce_vec<4>(buf, dgam, slab + a.o_ln_g, L);
    dvs_reduce_vec<4>(buf, dbet, slab + a.o_ln_b, L);
    // db2: scalar, lanes g == 0 hold partials
    {
        const float s = dvs_sum_wave(L.g == 0 ? db2 : 0.f);
        for (int w = 0; w < L.nwaves; ++w) {
            if (L.wave == w && L.lane == 0) buf[0] = (w == 0 ? 0.f : buf[0]) + s;
            __syncthreads();
        }
        if (threadIdx.x == 0) slab[a.o_edge2_b] = buf[0];
    }
}

void dvs_launch_loss_bwd(const LossArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = dvs_loss_lds_floats(4, 3) * 4;
    DVS_SET_LDS(k_loss_bwd, lds);
    DVS_LAUNCH(k_loss_bwd, dim3(grid), dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Embedding backward (autograd of pace.py:201-221, 1181-1184).  The one-hot first layers are row gathers forward,
// so their gradients are row scatters: per-wave LDS accumulators (ds_add_f32), summed over waves in fixed order.
// Handles up to two gradient sources per DAG (encoder-side and decoder-side embeddings: same weights, different
// dropout sites).
// ---------------------------------------------------------------------------------------------------------
constexpr int EMB_LDW2 = 36;
__global__ __launch_bounds__(256) void k_embed_bwd(EmbedArgs a, const float* gout2, int site2) {
    DVS_DYN_LDS(smem);
    const int N = a.dims.N, C = a.dims.C;
    float* W1 = (float*)smem;                        // [32][LD]
    float* W2 = W1 + 2 * DVS_MAXTOK * DVS_LD;        // [64][36]
    float* labw = W2 + 64 * EMB_LDW2;                // [32][16]
    float* labb = labw + 32 * 16;                    // [32]
    float* scr0 = labb + 32;                         // nwaves tiles
    float* accW1_0 = scr0 + 4 * DVS_SCR;             // nwaves x [32][64]
    float* accLab_0 = accW1_0 + 4 * 2048;            // nwaves x [32][16]
    dvs_stage_matrix(W1, DVS_LD, a.W1, 64, 2 * N, 64);
    dvs_stage_matrix(W2, EMB_LDW2, a.W2, 32, 64, 32);
    for (int i = threadIdx.x; i < 32 * 16; i += blockDim.x) {
        const int f = i >> 4, c = i & 15;
        labw[i] = c < C ? a.lab_w[f * C + c] : 0.f;
    }
    dvs_stage_vector(labb, a.lab_b, 32);
    for (int i = threadIdx.x; i < 4 * 2048 + 4 * 512; i += blockDim.x) accW1_0[i] = 0.f;
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    float* scr = scr0 + L.wave * DVS_SCR;
    float* accW1 = accW1_0 + L.wave * 2048;
    float* accLab = accLab_0 + L.wave * 512;
    f4 dW2[4][2], dlabb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) dW2[i][0] = dW2[i][1] = f4_zero();
    dlabb[0] = dlabb[1] = f4_zero();
    for (int dag = blockIdx.x * L.nwaves + L.wave; dag < a.dims.B; dag += gridDim.x * L.nwaves) {
        const DvsRecord* rec = a.rec + dag;
        const bool valid = L.r < N;
        const int label = rec->label[L.r];
        const int pos = rec->pos[L.r];
        const unsigned parents = rec->parents[L.r];
        const uint32_t gdag = a.dims.dag_offset + dag;
        // hidden of the positional encoder (post-relu, pre-dropout), as in the forward
        f4 e1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) e1[t] = *(const f4*)(W1 + pos * DVS_LD + 16 * t + 4 * L.g);
        for (int j = 0; j < N; ++j) {
            const int pj = rec->pos[j];
            const float on = ((parents >> j) & 1u) ? 1.f : 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t) e1[t] += *(const f4*)(W1 + (N + pj) * DVS_LD + 16 * t + 4 * L.g) * on;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) e1[t][kk] = valid ? fmaxf(e1[t][kk], 0.f) : 0.f;
        for (int src = 0; src < 2; ++src) {
            const float* gsrc = src == 0 ? a.gout : gout2;
            if (!gsrc) continue;
            const int site = src == 0 ? a.site : site2;
            f4 gx[4];
            dvs_load_grad(gx, gsrc, dag, N, L);
            // label half
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int f = 16 * t + 4 * L.g + kk;
                    const float pre = labw[f * 16 + label] + labb[f];
                    const float d = (valid && pre > 0.f) ? gx[t][kk] : 0.f;
                    dlabb[t][kk] += d;
                    if (valid) atomicAdd(&accLab[f * 16 + label], d);
                }
            // positional half
            f4 e1d[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) e1d[t] = e1[t];
            const uint32_t k1 = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, site, gdag);
            dvs_dropout_tile(e1d, k1, D, L);
            f4 de2[2];
            {
                f4 tmp[4] = {gx[2], gx[3], f4_zero(), f4_zero()};
                dvs_dropout_tile(tmp, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, site + 1, gdag), D, L);
                de2[0] = tmp[0];
                de2[1] = tmp[1];
            }
            f4 e1dN[4], de2N[2];
            dvs_t2n<4>(e1dN, e1d, scr, L);
            dvs_t2n<2>(de2N, de2, scr, L);
            dvs_outer_acc<4, 2>(dW2, e1dN, de2N);
            f4 de1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
            dvs_mat_T<4, 2>(de1, de2, W2, EMB_LDW2, 0, L);
            dvs_dropout_tile(de1, k1, D, L);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) de1[t][kk] = e1[t][kk] > 0.f ? de1[t][kk] : 0.f;
            if (valid) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) atomicAdd(&accW1[pos * 64 + 16 * t + 4 * L.g + kk], de1[t][kk]);
            }
            for (int j = 0; j < N; ++j) {
                if (valid && ((parents >> j) & 1u)) {
                    const int row = N + rec->pos[j];
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk) atomicAdd(&accW1[row * 64 + 16 * t + 4 * L.g + kk], de1[t][kk]);
                }
            }
        }
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    for (int i = threadIdx.x; i < 2 * N * 64; i += blockDim.x) {
        float s = 0.f;
        for (int w = 0; w < L.nwaves; ++w) s += accW1_0[w * 2048 + i];
        slab[a.oW1 + i] = s;
    }
    for (int i = threadIdx.x; i < 32 * C; i += blockDim.x) {
        const int f = i / C, c = i - f * C;
        float s = 0.f;
        for (int w = 0; w < L.nwaves; ++w) s += accLab_0[w * 512 + f * 16 + c];
        slab[a.olab_w + i] = s;
    }
    __syncthreads();
    float* buf = (float*)smem;
    dvs_reduce_dw<4, 2>(buf, dW2, slab + a.oW2, L);
    dvs_reduce_vec<2>(buf, dlabb, slab + a.olab_b, L);
}

void dvs_launch_embed_bwd(const EmbedArgs& a, const float* gout2, int site2, int grid, dvs_stream_t st) {
    const size_t lds = (2 * DVS_MAXTOK * DVS_LD + 64 * EMB_LDW2 + 32 * 16 + 32 + 4 * DVS_SCR + 4 * 2048 + 4 * 512) * 4;
    DVS_SET_LDS(k_embed_bwd, lds);
    DVS_LAUNCH(k_embed_bwd, dim3(grid), dim3(256), lds, st, a, gout2, site2);
}

// ---------------------------------------------------------------------------------------------------------
// Latent block backward, part 1 (one wave per 16 DAGs): dz^T = fc3^T dmem^T; through the reparameterisation and
// the KL term to (d mu, d logvar); then d enc_out^T = [fc1;fc2]^T [dmu;dlogvar]^T, stored frag order.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_latent_bwd(LatentBwdArgs a) {
    const Lane L = dvs_lane();
    const int B = a.dims.B, N = a.dims.N;
    const int ngroups = (B + 15) >> 4;
    const int ldw = N * 64;
    const float gkl = a.gcoef[1];
    for (int grp = blockIdx.x * L.nwaves + L.wave; grp < ngroups; grp += gridDim.x * L.nwaves) {
        const int dag = grp * 16 + L.r;
        const bool dvalid = dag < B;
        f4 dz[2] = {f4_zero(), f4_zero()};
        for (int m = 0; m < 64; ++m) {
            const int tok = 4 * (m & 3) + L.g;
            const int fb = 16 * (m >> 4) + 4 * ((m >> 2) & 3);
            const bool tv = tok < N;
            const f4 gb = dvalid ? *(const f4*)(a.gmem + (size_t)dag * DVS_TILE + 16 * m + 4 * L.g) : f4_zero();
            const float* wp = a.fc3_w + (size_t)((tv ? tok : 0) * 64 + fb) * 32 + L.r;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                f4 wa;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) wa[kk] = tv ? wp[kk * 32 + 16 * t] : 0.f;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dz[t] = dvs_mfma(wa[kk], gb[kk], dz[t]);
            }
        }
        // dz[t][reg] = d z[o = 16t + 4g + reg][dag r]
        f4 dout[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const size_t o4 = (size_t)(dvalid ? dag : 0) * 32 + 16 * t + 4 * L.g;
            const f4 mu = *(const f4*)(a.mu + o4), lv = *(const f4*)(a.logvar + o4), ev = *(const f4*)(a.epsv + o4);
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const float dzz = dvalid ? dz[t][reg] : 0.f;
                float dmu = dzz + gkl * mu[reg];
                float dlv = gkl * 0.5f * (__expf(lv[reg]) - 1.0f);
                if (a.dims.training) dlv += dzz * ev[reg] * 0.5f * __expf(0.5f * lv[reg]);
                dout[t][reg] = dvalid ? dmu : 0.f;
                dout[t + 2][reg] = dvalid ? dlv : 0.f;
            }
            if (dvalid) {
                *(f4*)(a.gz + (size_t)dag * 64 + 16 * t + 4 * L.g) = dout[t];
                *(f4*)(a.gz + (size_t)dag * 64 + 32 + 16 * t + 4 * L.g) = dout[t + 2];
            }
        }
        // d enc_out^T[k'][dag] = sum_o Wfc[o][col(k')] dout^T[o][dag]
        for (int m = 0; m < 64; ++m) {
            const int fb = 16 * (m >> 4) + 4 * ((m >> 2) & 3);
            const int tokD = 4 * (m & 3) + L.g;
            const int tokA = 4 * (m & 3) + (L.r >> 2);
            const bool av = tokA < N;
            const size_t colA = (size_t)(av ? tokA : 0) * 64 + fb + (L.r & 3);
            f4 o = f4_zero();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* wp = (t < 2 ? a.fc1_w : a.fc2_w) + (size_t)(16 * (t & 1) + 4 * L.g) * ldw + colA;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) o = dvs_mfma(av ? wp[(size_t)kk * ldw] : 0.f, dout[t][kk], o);
            }
            if (dvalid) *(f4*)(a.genc + (size_t)dag * DVS_TILE + 16 * m + 4 * L.g) = tokD < N ? o : f4_zero();
        }
    }
}

void dvs_launch_latent_bwd(const LatentBwdArgs& a, dvs_stream_t st) {
    const int ngroups = (a.dims.B + 15) / 16;
    DVS_LAUNCH(k_latent_bwd, dim3((ngroups + 3) / 4), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Latent block backward, part 2: weight gradients of fc1/fc2/fc3 — batch-contraction GEMMs.  Workgroup s owns the
// DAG range of slab s and writes its partial products straight into its slab (each wave a disjoint set of output
// tiles), so no cross-wave reduction is needed.
//   dWfc[o][col(k')] = sum_dag dout[dag][o] X[dag][k'] ;  dW3[row(k')][o] = sum_dag dmem[dag][k'] z[dag][o]
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_fc_dw(FcDwArgs a) {
    const Lane L = dvs_lane();
    const int B = a.dims.B, N = a.dims.N;
    const int ldw = N * 64;
    const int per = (B + a.nslab - 1) / a.nslab;
    const int d0 = blockIdx.x * per;
    const int d1 = (d0 + per < B) ? d0 + per : B;
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    // ---- biases (thread per element) ----------------------------------------------------------------------
    for (int i = threadIdx.x; i < 64; i += blockDim.x) {
        float s = 0.f;
        for (int d = d0; d < d1; ++d) s += a.gz[(size_t)d * 64 + i];
        if (i < 32) slab[a.o_fc1_b + i] = s; else slab[a.o_fc2_b + i - 32] = s;
    }
    for (int i = threadIdx.x; i < N * 64; i += blockDim.x) {
        const int tok = i >> 6, f = i & 63;
        const int k = (f >> 4) * 256 + ((((f >> 2) & 3) * 16 + tok) << 2) + (f & 3);   // frag index of (tok, f)
        float s = 0.f;
        for (int d = d0; d < d1; ++d) s += a.gmem[(size_t)d * DVS_TILE + k];
        slab[a.o_fc3_b + i] = s;
    }
    // ---- weights: each wave walks output column tiles m = wave, wave+4, ... ------------------------------------
    for (int m = L.wave; m < 64; m += L.nwaves) {
        f4 acc[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};   // dWfc tiles (o tile t) x k' tile m
        f4 acc3[2] = {f4_zero(), f4_zero()};                           // dW3 tile: rows k' tile m, cols o tile t
        for (int c0 = d0; c0 < d1; c0 += 16) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int d = c0 + 4 * ks + L.g;
                const bool dv = d < d1;
                const size_t dd = dv ? d : d0;
                const float xb = dv ? a.xenc[dd * DVS_TILE + 16 * m + L.r] : 0.f;
                const float gm = dv ? a.gmem[dd * DVS_TILE + 16 * m + L.r] : 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float ga = dv ? a.gz[dd * 64 + 16 * t + L.r] : 0.f;
                    acc[t] = dvs_mfma(ga, xb, acc[t]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const float zb = dv ? a.z[dd * 32 + 16 * t + L.r] : 0.f;
                    acc3[t] = dvs_mfma(gm, zb, acc3[t]);
                }
            }
        }
        // acc[t][reg] = dWfc[o = 16t + 4g + reg][k' = 16m + r]
        {
            const int tok = 4 * (m & 3) + (L.r >> 2);
            const int f = 16 * (m >> 4) + 4 * ((m >> 2) & 3) + (L.r & 3);
            if (tok < N) {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int o = 16 * (t & 1) + 4 * L.g + reg;
                        slab[(t < 2 ? a.o_fc1_w : a.o_fc2_w) + (size_t)o * ldw + tok * 64 + f] = acc[t][reg];
                    }
            }
        }
        // acc3[t][reg] = dW3[row(k' = 16m + 4g + reg)][o = 16t + r]
        {
            const int tok = 4 * (m & 3) + L.g;
            const int fb = 16 * (m >> 4) + 4 * ((m >> 2) & 3);
            if (tok < N) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        slab[a.o_fc3_w + (size_t)(tok * 64 + fb + reg) * 32 + 16 * t + L.r] = acc3[t][reg];
            }
        }
    }
}

void dvs_launch_fc_dw(const FcDwArgs& a, dvs_stream_t st) {
    DVS_LAUNCH(k_fc_dw, dim3(a.nslab), dim3(256), 0, st, a);
}
