// Attention-core backward: from d(pre) of an attention sublayer to d(q), d(k), d(v) projections and dWo/dbo.
//
// Recomputes q,k,v and the probabilities of every head (nothing but the sublayer input was saved).  Two
// orientations of the 16x16 score tile are used so that every product contracts over the MFMA row index of a
// register-resident operand (see dvs_device.h):
//   "T":  P^T[j=4g+reg][i=r]  -> softmax statistics (in-lane + 2 shuffles), dP^T, dS^T -> dq^T
//   "S":  P  [i=4g+reg][j=r]  -> recomputed from the T statistics (3 shuffles), dP, dS -> dk^T, dv^T
// q,k (T-layout) feed the score products directly; their N-layout copies (for dq/dk), v^T and dO (N) come from
// per-wave LDS transposes.
#include "dvs_backward.h"

struct AttnBLds {
    float *Win, *Wout, *inb, *outb, *lg, *lb, *scr;
};
__device__ __forceinline__ AttnBLds attnb_lds(char* smem) {
    AttnBLds l;
    l.Win = (float*)smem;
    l.Wout = l.Win + 192 * DVS_LD;
    l.inb = l.Wout + 64 * DVS_LD;
    l.outb = l.inb + 192;
    l.lg = l.outb + 64;
    l.lb = l.lg + 64;
    l.scr = l.lb + 64;
    return l;
}
static size_t attnb_lds_floats(int nwaves) { return 256 * DVS_LD + 192 + 64 + 128 + (size_t)nwaves * DVS_SCR; }

__device__ __forceinline__ f4 drop_T(f4 p, uint32_t key, int h, const DvsDrop& D, const Lane& L) {
    if (!D.on) return p;
    const uint32_t p0 = (uint32_t)((h * 16 + L.r) * 8 + 2 * L.g);
    const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
    p[0] = ((h0 & 0xFFFFu) >= D.thr16) ? p[0] * D.scale : 0.f;
    p[1] = ((h0 >> 16) >= D.thr16) ? p[1] * D.scale : 0.f;
    p[2] = ((h1 & 0xFFFFu) >= D.thr16) ? p[2] * D.scale : 0.f;
    p[3] = ((h1 >> 16) >= D.thr16) ? p[3] * D.scale : 0.f;
    return p;
}
// same mask in the S orientation: register reg holds (i = 4g+reg, j = r) -> element ((h*16+i)*16 + j)
__device__ __forceinline__ f4 drop_S(f4 p, uint32_t key, int h, const DvsDrop& D, const Lane& L) {
    if (!D.on) return p;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const uint32_t e = (uint32_t)((h * 16 + 4 * L.g + reg) * 16 + L.r);
        p[reg] = dvs_dropout_elem(p[reg], key, e, D);
    }
    return p;
}

__global__ __launch_bounds__(256) void k_attn_bwd(AttnBwdArgs a) {
    DVS_DYN_LDS(smem);
    const AttnBLds l = attnb_lds(smem);
    dvs_stage_matrix(l.Win, DVS_LD, a.in_w, 64, 192, 64);
    dvs_stage_matrix(l.Wout, DVS_LD, a.out_w, 64, 64, 64);
    dvs_stage_vector(l.inb, a.in_b, 192);
    dvs_stage_vector(l.outb, a.out_b, 64);
    if (a.ln.stats) {
        dvs_stage_vector(l.lg, a.ln.g, 64);
        dvs_stage_vector(l.lb, a.ln.b, 64);
    }
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N;
    float* scr = l.scr + L.wave * DVS_SCR;
    const float scale = 0.35355339059327373f;
    f4 dWo[4][4], dbo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dbo[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) dWo[i][j] = f4_zero();
    }
    for (int dag = blockIdx.x * L.nwaves + L.wave; dag < a.dims.B; dag += gridDim.x * L.nwaves) {
        const uint32_t gdag = a.dims.dag_offset + dag;
        const uint32_t kprob = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_prob, gdag);
        f4 q[4], k[4], v[4];
        {
            f4 x[4], kv[4], dummy[4];
            float rstd;
            dvs_load_x<false>(x, dummy, rstd, a.xin, a.ln, l.lg, l.lb, dag, N, L);
            if (a.kv) {
                dvs_load_tile(kv, a.kv, dag, L);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) kv[t] = x[t];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                q[t] = dvs_vecT(l.inb, t, L);
                k[t] = dvs_vecT(l.inb + 64, t, L);
                v[t] = f4_splat(l.inb[128 + 16 * t + L.r]);
            }
            dvs_mat_T<4, 4>(q, x, l.Win, DVS_LD, 0, L);
            dvs_mat_T<4, 4>(k, kv, l.Win, DVS_LD, 64, L);
            dvs_mat_N<4, 4>(v, kv, l.Win, DVS_LD, 128, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] *= scale;
        }
        f4 dy[4];
        dvs_load_grad(dy, a.gpre, dag, N, L);
        dvs_dropout_tile(dy, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L);
#pragma unroll
        for (int t = 0; t < 4; ++t) dbo[t] += dy[t];
        f4 qN[4], kN[4], vT[4], dyN[4], dOT[4], dON[4];
        dvs_t2n<4>(qN, q, scr, L);
        dvs_t2n<4>(kN, k, scr, L);
        dvs_n2t<4>(vT, v, scr, L);
        dvs_t2n<4>(dyN, dy, scr, L);
#pragma unroll
        for (int t = 0; t < 4; ++t) dOT[t] = f4_zero();
        dvs_mat_Tt<4, 4>(dOT, dy, l.Wout, DVS_LD, 0, L);
        dvs_t2n<4>(dON, dOT, scr, L);

        const unsigned allowed_r = a.rec[dag].allowed[L.r];
        unsigned al4[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) al4[reg] = (unsigned)__shfl((int)allowed_r, 4 * L.g + reg);

        f4 oN[4], dq[4], dk[4], dv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) oN[t] = dq[t] = dk[t] = dv[t] = f4_zero();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int hs = 0; hs < 2; ++hs) {
                const int h = 2 * t + hs;
                const bool mine_g = (L.g >> 1) == hs, mine_r = (L.r >> 3) == hs;
                // ---- T orientation ------------------------------------------------------------------------
                f4 s = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) s = dvs_mfma(mine_g ? k[t][kk] : 0.f, q[t][kk], s);
                float mx = -3.0e38f;
                bool ok[4];
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    ok[reg] = (allowed_r >> (4 * L.g + reg)) & 1u;
                    mx = ok[reg] ? fmaxf(mx, s[reg]) : mx;
                }
                const float m = dvs_max_g(mx);
                f4 pT;
                float sum = 0.f;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    pT[reg] = ok[reg] ? __expf(s[reg] - m) : 0.f;
                    sum += pT[reg];
                }
                const float den = dvs_sum_g(sum);
                pT *= (1.0f / den);
                const f4 pdT = drop_T(pT, kprob, h, D, L);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) oN[t] = dvs_mfma(pdT[kk], mine_r ? v[t][kk] : 0.f, oN[t]);
                f4 dpT = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dpT = dvs_mfma(mine_g ? vT[t][kk] : 0.f, dOT[t][kk], dpT);
                dpT = drop_T(dpT, kprob, h, D, L);
                float dl = 0.f;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dl += pT[reg] * dpT[reg];
                const float delta = dvs_sum_g(dl);
                f4 dsT;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) dsT[reg] = pT[reg] * (dpT[reg] - delta);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dq[t] = dvs_mfma(mine_r ? kN[t][kk] : 0.f, dsT[kk], dq[t]);
                // ---- S orientation ------------------------------------------------------------------------
                f4 s2 = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) s2 = dvs_mfma(mine_g ? q[t][kk] : 0.f, k[t][kk], s2);
                f4 p, dp = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dp = dvs_mfma(mine_g ? dOT[t][kk] : 0.f, vT[t][kk], dp);
                dp = drop_S(dp, kprob, h, D, L);
                f4 ds;
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const int src = 4 * L.g + reg;
                    const float m_i = __shfl(m, src), den_i = __shfl(den, src), delta_i = __shfl(delta, src);
                    const bool oki = (al4[reg] >> L.r) & 1u;
                    p[reg] = oki ? __expf(s2[reg] - m_i) / den_i : 0.f;
                    ds[reg] = p[reg] * (dp[reg] - delta_i);
                }
                const f4 pd = drop_S(p, kprob, h, D, L);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dk[t] = dvs_mfma(mine_r ? qN[t][kk] : 0.f, ds[kk], dk[t]);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) dv[t] = dvs_mfma(mine_r ? dON[t][kk] : 0.f, pd[kk], dv[t]);
            }
        }
        dvs_outer_acc<4, 4>(dWo, dyN, oN);
#pragma unroll
        for (int t = 0; t < 4; ++t) dq[t] *= scale;
        dvs_store_tile(a.gq, dag, dq, L);
        dvs_store_tile(a.gk, dag, dk, L);
        dvs_store_tile(a.gv, dag, dv, L);
    }
    __syncthreads();
    float* buf = (float*)smem;
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    dvs_reduce_dw<4, 4>(buf, dWo, slab + a.o_out_w, L);
    dvs_reduce_vec<4>(buf, dbo, slab + a.o_out_b, L);
}

void dvs_launch_attn_bwd(const AttnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = attnb_lds_floats(4) * 4;
    DVS_SET_LDS(k_attn_bwd, lds);
    DVS_LAUNCH(k_attn_bwd, dim3(grid), dim3(256), lds, st, a);
}
