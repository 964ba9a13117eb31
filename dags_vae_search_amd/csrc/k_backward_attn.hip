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
#include "dvs_wimg.h"

struct AttnBLds {
    float *inb, *outb, *lg, *lb, *slots, *stats;
    // bf16x3 images (dvs_bf16.h): the in-projection rows (q, k, v are recomputed through a softmax — smooth, so their
    // ~1e-5 perturbation stays a ~1e-5 perturbation of the gradient; the FFN's hidden, whose SIGN is a mask, is recomputed
    // in exact fp32 instead) and Wo^T (dO^T = Wo^T dy^T, a pure gradient product)
    dvs_bf16 *Winh, *Winl, *WoTh, *WoTl;
    int* gcount;
};
__device__ __forceinline__ AttnBLds attnb_lds(char* smem) {
    AttnBLds l;
    l.Winh = (dvs_bf16*)smem;
    l.Winl = l.Winh + 192 * DVS_LDB;
    l.WoTh = l.Winl + 192 * DVS_LDB;
    l.WoTl = l.WoTh + 64 * DVS_LDB;
    l.inb = (float*)(l.WoTl + 64 * DVS_LDB);
    l.outb = l.inb + 192;
    l.lg = l.outb + 64;
    l.lb = l.lg + 64;
    l.slots = l.lb + 64;                       // per wave: A (d y, row-major) and B (transpose scratch, then O)
    l.stats = l.slots + 8 * 2 * DVS_SCR;       // per wave 128 floats: lse / delta exchange
    l.gcount = (int*)(l.stats + 8 * 128);
    return l;
}
static size_t attnb_lds_floats() {
    return 512 * DVS_LDB / 2 + 192 + 64 + 128 + (size_t)8 * 2 * DVS_SCR + 8 * 128 + 16;
}

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

// 8 waves per workgroup in two independent groups of four (dvs_backward.h); one DAG per wave per iteration.  The
// out-projection gradient is accumulated cooperatively from the parked d y and O tiles, d q / d k / d v tiles are stored
// as soon as their head pair is finished, so a wave stays within 256 registers and two waves share each SIMD.
__global__ __launch_bounds__(512) void k_attn_bwd(AttnBwdArgs a) {
    DVS_DYN_LDS(smem);
    const AttnBLds l = attnb_lds(smem);
    dvs_copy_image(l.Winh, (const dvs_bf16*)a.wimg + DvsAttnImg::Win, 2 * 192 * DVS_LDB);   // parts hi, mid of the x6 triple = the x3 pair
    dvs_copy_image(l.WoTh, (const dvs_bf16*)a.wimg + DvsAttnImg::WoutT, (int)(2 * DVS_IMG64));
    dvs_stage_vector_perm(l.inb, a.in_b, 192);
    dvs_stage_vector(l.outb, a.out_b, 64);
    if (a.ln.stats) {
        dvs_stage_vector(l.lg, a.ln.g, 64);
        dvs_stage_vector(l.lb, a.ln.b, 64);
    }
    if (threadIdx.x < 2) l.gcount[threadIdx.x] = 0;
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N, B = a.dims.B;
    float* sA = l.slots + L.wave * 2 * DVS_SCR;
    float* sB = sA + DVS_SCR;
    float* st = l.stats + L.wave * 128;
    DvsGroup G = {l.gcount + (L.wave >> 2), 0};
    const float scale = 0.35355339059327373f;
    f4 aWo[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
    float vbo = 0.f;
    for (int base = blockIdx.x * 8; base < B; base += gridDim.x * 8) {
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
            dvs_load_x<false>(x, dummy, rstd, a.xin, a.ln, l.lg, l.lb, dg, Nl, L);
            if (a.kv) {
                dvs_load_tile(kv, a.kv, dg, L);
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
            dvs_park_T(sA, dy, L);                     // stays parked until the cooperative dWo below
            dvs_wave_sync();
            vbo += dvs_colsum(sA, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) dOT[t] = f4_zero();
            dvs_matb_T<4>(dOT, dvs_split_T(dy), l.WoTh, l.WoTl, 0, L);
        }
        dvs_t2n<4>(qN, q, sB, L);
        dvs_t2n<4>(kN, k, sB, L);
        dvs_n2t<4>(vT, v, sB, L);
        dvs_t2n<4>(dON, dOT, sB, L);

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
            f4 pT[2], dsT[2];
            float lse[2], delta[2];
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
                for (int u = 0; u < 2; ++u) m[u] = fmaxf(m[u], __shfl_xor(m[u], 16));
#pragma unroll
                for (int u = 0; u < 2; ++u) m[u] = fmaxf(m[u], __shfl_xor(m[u], 32));
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
                for (int u = 0; u < 2; ++u) den[u] += __shfl_xor(den[u], 16);
#pragma unroll
                for (int u = 0; u < 2; ++u) den[u] += __shfl_xor(den[u], 32);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    pT[u] *= (1.0f / den[u]);
                    lse[u] = m[u] + __logf(den[u]);
                }
            }
            {
                f4 mk[2], dpT[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    mk[u] = mask_T(kprob, 2 * t + u, D, L);
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
                    float* po = sB + (4 * L.g) * DVS_LD + 16 * t + L.r;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) po[reg * DVS_LD] = hsel ? ob[reg] : oa[reg];
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
                for (int u = 0; u < 2; ++u) delta[u] += __shfl_xor(delta[u], 16);
#pragma unroll
                for (int u = 0; u < 2; ++u) delta[u] += __shfl_xor(delta[u], 32);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) dsT[u][reg] = pT[u][reg] * (dpT[u][reg] - delta[u]);
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
            // row statistics (lse, delta) move from lanes r = i to the S-orientation registers i = 4g+reg
            if (L.g == 0) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    st[u * 32 + L.r] = lse[u];
                    st[u * 32 + 16 + L.r] = delta[u];
                }
            }
            dvs_wave_sync();
            // ---- S orientation: reg <-> (query i = 4g+reg, key j = r) --------------------------------------------
            {
                f4 s2[2] = {f4_zero(), f4_zero()}, dp[2] = {f4_zero(), f4_zero()};
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    s2[0] = dvs_mfma(q[t][kk], k[t][kk], s2[0]);
                    s2[1] = dvs_mfma(q[t][kk + 2], k[t][kk + 2], s2[1]);
                    dp[0] = dvs_mfma(dOT[t][kk], vT[t][kk], dp[0]);
                    dp[1] = dvs_mfma(dOT[t][kk + 2], vT[t][kk + 2], dp[1]);
                }
                f4 ds[2], pd[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const f4 lse_i = *(const f4*)(st + u * 32 + 4 * L.g);
                    const f4 del_i = *(const f4*)(st + u * 32 + 16 + 4 * L.g);
                    const f4 mk = mask_S(kprob, 2 * t + u, D, L);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const bool oki = (al4[reg] >> L.r) & 1u;
                        const float p = oki ? __expf(s2[u][reg] - lse_i[reg]) : 0.f;
                        ds[u][reg] = p * (dp[u][reg] * mk[reg] - del_i[reg]);
                        pd[u][reg] = p * mk[reg];
                    }
                }
                dvs_wave_sync();
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
        dvs_coop_dw(aWo, l.slots, l.slots + DVS_SCR, 2 * DVS_SCR, L);
        dvs_group_barrier(G, L);
    }
    __syncthreads();
    float* slab = a.slab + (size_t)blockIdx.x * a.P;
    dvs_coop_store((float*)smem, slab + a.o_out_w, aWo, L, false, true);     // columns back to parameter order
    float* red = (float*)smem;
    red[L.wave * 64 + L.lane] = vbo;
    __syncthreads();
    if (threadIdx.x < 64) {
        float s = 0.f;
        for (int w = 0; w < 8; ++w) s += red[w * 64 + threadIdx.x];
        slab[a.o_out_b + threadIdx.x] = s;
    }
}

void dvs_launch_attn_bwd(const AttnBwdArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = attnb_lds_floats() * 4;
    DVS_SET_LDS(k_attn_bwd, lds);
    DVS_LAUNCH(k_attn_bwd, dim3(grid), dim3(512), lds, st, a);
}
