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
static size_t attnb_lds_floats(int nwaves) { return 256 * DVS_LD + 192 + 64 + 128 + (size_t)nwaves * (DVS_SCR + 3 * DVS_TILE); }

// dropout multipliers (0 or 1/keep) of head h; T orientation: reg <-> (i = r, j = 4g+reg); S orientation:
// reg <-> (i = 4g+reg, j = r); element index ((h*16 + i)*16 + j) in both.
__device__ __forceinline__ f4 mask_T(uint32_t key, int h, const DvsDrop& D, const Lane& L) {
    if (!D.on) return f4_splat(1.f);
    const uint32_t p0 = (uint32_t)((h * 16 + L.r) * 8 + 2 * L.g);
    const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
    f4 m;
    m[0] = ((h0 & 0xFFFFu) >= D.thr16) ? D.scale : 0.f;
    m[1] = ((h0 >> 16) >= D.thr16) ? D.scale : 0.f;
    m[2] = ((h1 & 0xFFFFu) >= D.thr16) ? D.scale : 0.f;
    m[3] = ((h1 >> 16) >= D.thr16) ? D.scale : 0.f;
    return m;
}
__device__ __forceinline__ f4 mask_S(uint32_t key, int h, const DvsDrop& D, const Lane& L) {
    if (!D.on) return f4_splat(1.f);
    f4 m;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
        const uint32_t e = (uint32_t)((h * 16 + 4 * L.g + reg) * 16 + L.r);
        m[reg] = dvs_dropout_elem(1.0f, key, e, D);
    }
    return m;
}

__global__ __launch_bounds__(256) void k_attn_bwd(AttnBwdArgs a) {
    DVS_DYN_LDS(smem);
    const AttnBLds l = attnb_lds(smem);
    dvs_stage_matrix_perm(l.Win, DVS_LD, a.in_w, 64, 192, 64, true, false);    // head-aligned slot order (dvs_device.h)
    dvs_stage_matrix_perm(l.Wout, DVS_LD, a.out_w, 64, 64, 64, false, true);
    dvs_stage_vector_perm(l.inb, a.in_b, 192);
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
    float* pf = l.scr + L.nwaves * DVS_SCR + L.wave * 3 * DVS_TILE;   // LDS-DMA landing zone: x, kv, d pre
    const float scale = 0.35355339059327373f;
    f4 dWo[4][4], dbo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        dbo[i] = f4_zero();
#pragma unroll
        for (int j = 0; j < 4; ++j) dWo[i][j] = f4_zero();
    }
    const int stride = gridDim.x * L.nwaves;
    int dag = blockIdx.x * L.nwaves + L.wave;
    auto request = [&](int d) {
        dvs_prefetch_tile(pf, a.xin, d, L);
        if (a.kv) dvs_prefetch_tile(pf + DVS_TILE, a.kv, d, L);
        dvs_prefetch_tile(pf + 2 * DVS_TILE, a.gpre, d, L);
    };
    if (dag < a.dims.B) request(dag);
    for (; dag < a.dims.B; dag += stride) {
        dvs_prefetch_wait();
        const uint32_t gdag = a.dims.dag_offset + dag;
        const uint32_t kprob = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_prob, gdag);
        f4 q[4], k[4], v[4], dy[4];
        dvs_load_grad(dy, a.gpre, dag, N, L, pf + 2 * DVS_TILE);
        const unsigned allowed_r = a.rec[dag].allowed[L.r];
        {
            f4 x[4], kv[4], dummy[4];
            float rstd;
            dvs_load_x<false>(x, dummy, rstd, a.xin, a.ln, l.lg, l.lb, dag, N, L, pf);
            if (a.kv) dvs_slot_tile(kv, pf + DVS_TILE, L);
            dvs_slot_release();
            if (dag + stride < a.dims.B) request(dag + stride);
            if (a.kv) {
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

        const bool hsel = ((L.r & 3) >> 1) != 0;     // N-layout lane r holds slot r = feature 4(r&3) + (r>>2): head bit
        bool ok[4];
        unsigned al4[4];
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            ok[reg] = (allowed_r >> (4 * L.g + reg)) & 1u;
            al4[reg] = (unsigned)__shfl((int)allowed_r, 4 * L.g + reg);
        }
        f4 oN[4], dq[4], dk[4], dv[4];
        // Heads are processed four at a time (tiles 2*half, 2*half+1): enough independent chains to cover the MFMA and
        // cross-lane latencies, half the temporaries of an all-heads pass (the kernel sits at the 512-register limit).
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // ---- T orientation: reg <-> (query i = r, key j = 4g+reg) --------------------------------------------
            f4 pT[4], dsT[4];
            float lse[4], delta[4];
            {
                f4 sT[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) sT[u] = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 2 * half + tt;
                        sT[2 * tt] = dvs_mfma(k[t][kk], q[t][kk], sT[2 * tt]);
                        sT[2 * tt + 1] = dvs_mfma(k[t][kk + 2], q[t][kk + 2], sT[2 * tt + 1]);
                    }
                float m[4], den[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float mx = -3.0e38f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) mx = ok[reg] ? fmaxf(mx, sT[u][reg]) : mx;
                    m[u] = mx;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = fmaxf(m[u], __shfl_xor(m[u], 16));
#pragma unroll
                for (int u = 0; u < 4; ++u) m[u] = fmaxf(m[u], __shfl_xor(m[u], 32));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float sum = 0.f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        pT[u][reg] = ok[reg] ? __expf(sT[u][reg] - m[u]) : 0.f;
                        sum += pT[u][reg];
                    }
                    den[u] = sum;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) den[u] += __shfl_xor(den[u], 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) den[u] += __shfl_xor(den[u], 32);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    pT[u] *= (1.0f / den[u]);
                    lse[u] = m[u] + __logf(den[u]);
                }
            }
            {
                f4 mk[4], dpT[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mk[u] = mask_T(kprob, 4 * half + u, D, L);
                    dpT[u] = f4_zero();
                }
                // O (N-layout, for dWo) = P' V: both heads of a tile on all 16 slot columns, per-lane head select
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * half + tt;
                    f4 oa = f4_zero(), ob = f4_zero();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        oa = dvs_mfma(pT[2 * tt][kk] * mk[2 * tt][kk], v[t][kk], oa);
                        ob = dvs_mfma(pT[2 * tt + 1][kk] * mk[2 * tt + 1][kk], v[t][kk], ob);
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) oN[t][reg] = hsel ? ob[reg] : oa[reg];
                }
                // dP^T = V dO^T
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 2 * half + tt;
                        dpT[2 * tt] = dvs_mfma(vT[t][kk], dOT[t][kk], dpT[2 * tt]);
                        dpT[2 * tt + 1] = dvs_mfma(vT[t][kk + 2], dOT[t][kk + 2], dpT[2 * tt + 1]);
                    }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    dpT[u] *= mk[u];
                    float dl = 0.f;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) dl += pT[u][reg] * dpT[u][reg];
                    delta[u] = dl;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) delta[u] += __shfl_xor(delta[u], 16);
#pragma unroll
                for (int u = 0; u < 4; ++u) delta[u] += __shfl_xor(delta[u], 32);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) dsT[u][reg] = pT[u][reg] * (dpT[u][reg] - delta[u]);
            }
            // dq^T = K^T dS^T: all 16 slot rows per head, merged by register (rows reg 0,1 <-> first head of the tile)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int t = 2 * half + tt;
                f4 qa = f4_zero(), qb = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    qa = dvs_mfma(kN[t][kk], dsT[2 * tt][kk], qa);
                    qb = dvs_mfma(kN[t][kk], dsT[2 * tt + 1][kk], qb);
                }
                dq[t] = f4{qa[0], qa[1], qb[2], qb[3]};
            }
            // row statistics (lse, delta) of query i move from lanes r = i to the S-orientation registers i = 4g+reg
            // through the wave's scratch tile: one b128 read per head and quantity
            if (L.g == 0) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    scr[u * 32 + L.r] = lse[u];
                    scr[u * 32 + 16 + L.r] = delta[u];
                }
            }
            dvs_wave_sync();
            // ---- S orientation: reg <-> (query i = 4g+reg, key j = r) --------------------------------------------
            {
                f4 s2[4], dp[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) s2[u] = dp[u] = f4_zero();
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int t = 2 * half + tt;
                        s2[2 * tt] = dvs_mfma(q[t][kk], k[t][kk], s2[2 * tt]);
                        s2[2 * tt + 1] = dvs_mfma(q[t][kk + 2], k[t][kk + 2], s2[2 * tt + 1]);
                        dp[2 * tt] = dvs_mfma(dOT[t][kk], vT[t][kk], dp[2 * tt]);
                        dp[2 * tt + 1] = dvs_mfma(dOT[t][kk + 2], vT[t][kk + 2], dp[2 * tt + 1]);
                    }
                f4 ds[4], pd[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f4 lse_i = *(const f4*)(scr + u * 32 + 4 * L.g);
                    const f4 del_i = *(const f4*)(scr + u * 32 + 16 + 4 * L.g);
                    const f4 mk = mask_S(kprob, 4 * half + u, D, L);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const bool oki = (al4[reg] >> L.r) & 1u;
                        const float p = oki ? __expf(s2[u][reg] - lse_i[reg]) : 0.f;
                        ds[u][reg] = p * (dp[u][reg] * mk[reg] - del_i[reg]);
                        pd[u][reg] = p * mk[reg];
                    }
                }
                dvs_wave_sync();
                // dk^T = Q^T dS ;  dv^T = dO^T P'  (per head on all slot rows, merged by register)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int t = 2 * half + tt;
                    f4 ka = f4_zero(), kb = f4_zero(), va = f4_zero(), vb = f4_zero();
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {
                        ka = dvs_mfma(qN[t][kk], ds[2 * tt][kk], ka);
                        kb = dvs_mfma(qN[t][kk], ds[2 * tt + 1][kk], kb);
                        va = dvs_mfma(dON[t][kk], pd[2 * tt][kk], va);
                        vb = dvs_mfma(dON[t][kk], pd[2 * tt + 1][kk], vb);
                    }
                    dk[t] = f4{ka[0], ka[1], kb[2], kb[3]};
                    dv[t] = f4{va[0], va[1], vb[2], vb[3]};
                }
            }
            DVS_SCHED_FENCE();
        }
        dvs_outer_acc<4, 4>(dWo, dyN, oN);
#pragma unroll
        for (int t = 0; t < 4; ++t) dq[t] *= scale;
        dvs_store_tile(a.gq, dag, dq, L);
        dvs_store_tile(a.gk, dag, dk, L);
        dvs_store_tile(a.gv, dag, dv, L);
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    float* rW = (float*)smem;
    float* rv = rW + DVS_RED_MAT;
    dvs_stage_dw<4, 4>(rW, dWo, L);
    dvs_stage_vec<4>(rv, dbo, rv + DVS_RED_VEC + L.wave * DVS_SCR, L);
    __syncthreads();
    dvs_flush_dw<4, 4>(rW, slab + a.o_out_w, L, 64, 64, 64, false, true);   // columns back to parameter order
    dvs_flush_vec<4>(rv, slab + a.o_out_b, L);
}

void dvs_launch_attn_bwd(const AttnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = attnb_lds_floats(4) * 4;
    DVS_SET_LDS(k_attn_bwd, lds);
    DVS_LAUNCH(k_attn_bwd, dim3(grid), dim3(256), lds, st, a);
}
