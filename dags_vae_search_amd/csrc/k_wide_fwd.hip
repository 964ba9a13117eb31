// Wide path (17..48 tokens per DAG), forward kernels: records, embedding, attention sublayer, loss head.
// See dvs_wide.h for the execution model (workgroup = DAG, wave = 16-token tile, tiles meet in LDS).
#include "dvs_wide.h"

// ---------------------------------------------------------------------------------------------------------
// dvs_pack_features, wide records: one thread per (DAG, token slot).  Same checks as k_pack (k_forward.hip).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack_w(PackArgs a) {
    const int N = a.N, C = a.C;
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const int dag = (int)(gi / DVS_WTOK), i = (int)(gi % DVS_WTOK);
    if (dag >= a.B) return;
    int bad = 0, label = 0, pos = 0;
    uint64_t parents = 0, allowed = 1ull << i;
    if (i < N) {
        const float* lr = a.lab1h + ((size_t)dag * N + i) * C;
        int ones = 0;
        for (int c = 0; c < C; ++c) {
            const float v = lr[c];
            if (v == 1.0f) { label = c; ++ones; } else if (v != 0.0f) bad |= 1;
        }
        if (ones != 1) bad |= 1;
        const float* pr = a.pos1h + ((size_t)dag * N + i) * N;
        ones = 0;
        for (int c = 0; c < N; ++c) {
            const float v = pr[c];
            if (v == 1.0f) { pos = c; ++ones; } else if (v != 0.0f) bad |= 1;
        }
        if (ones != 1) bad |= 1;
        const float* ad = a.adj + (size_t)dag * N * N;
        for (int j = 0; j < N; ++j)
            if (ad[j * N + i] != 0.0f) parents |= 1ull << j;
        allowed = 0;
        const uint8_t* m0 = a.tmask + ((size_t)dag * 8 * N + i) * N;
        for (int j = 0; j < N; ++j)
            if (!m0[j]) allowed |= 1ull << j;
        for (int h = 1; h < 8; ++h) {
            const uint8_t* mh = a.tmask + (((size_t)dag * 8 + h) * N + i) * N;
            for (int j = 0; j < N; ++j)
                if ((mh[j] != 0) != (m0[j] != 0)) bad |= 2;
        }
        if (!((allowed >> i) & 1ull)) bad |= 4;
    }
    DvsRecordW* r = (DvsRecordW*)a.rec + dag;
    r->label[i] = (uint8_t)label;
    r->pos[i] = (uint8_t)pos;
    r->parents[i] = parents;
    r->allowed[i] = allowed;
    if (bad) atomicOr(a.status, bad);
}

void dvs_launch_pack_w(const PackArgs& a, dvs_stream_t st) {
    const long long n = (long long)a.B * DVS_WTOK;
    DVS_LAUNCH(k_pack_w, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// dvs_build_records, wide: row codec -> records, one thread per DAG on 64-bit rows (k_build_records restated for
// N <= 48; same PACE wrapping pace.py:1250-1288, FIFO-Kahn positions with the positions[v] = order[v] quirk
// pace.py:1245-1248/1286, ancestor closure pace.py:1307-1343).
// ---------------------------------------------------------------------------------------------------------
constexpr int BW_T = 64;
__global__ __launch_bounds__(BW_T) void k_build_records_w(BuildWArgs a) {
    __shared__ uint64_t s_child[DVS_WTOK][BW_T];
    __shared__ uint64_t s_reach[DVS_WTOK][BW_T];
    __shared__ unsigned char s_indeg[DVS_WTOK][BW_T];
    __shared__ unsigned char s_order[DVS_WTOK][BW_T];
    const int tid = threadIdx.x;
    const int dag = blockIdx.x * BW_T + tid;
    if (dag >= a.B) return;
    const int N = a.N, n = N - 3, out_id = N - 1;
    int bad = 0;
    for (int v = 0; v < DVS_WTOK; ++v) s_child[v][tid] = 0;
    s_child[0][tid] = 1ull << 1;
    uint64_t haspred = 0;
    for (int v = 0; v < n; ++v) {
        const uint64_t p = a.preds[(size_t)dag * n + v];
        if (p >> v) bad |= 8;                                   // an edge u -> v needs u < v
        const uint64_t pm = p & ((1ull << v) - 1ull);
        uint64_t c = pm;
        while (c) {
            const int u = dvs_ctz64(c);
            c &= c - 1;
            s_child[u + 2][tid] |= 1ull << (v + 2);
        }
        if (pm) haspred |= 1ull << v;
    }
    for (int v = 0; v < n; ++v)
        if (!((haspred >> v) & 1ull)) s_child[1][tid] |= 1ull << (v + 2);
    for (int v = 0; v < N - 1; ++v)
        if (s_child[v][tid] == 0) s_child[v][tid] = 1ull << out_id;
    for (int v = 0; v < DVS_WTOK; ++v) s_indeg[v][tid] = 0;
    for (int u = 0; u < N; ++u) {
        uint64_t c = s_child[u][tid];
        while (c) {
            const int v = dvs_ctz64(c);
            c &= c - 1;
            s_indeg[v][tid]++;
        }
    }
    int tail = 0;
    for (int v = 0; v < N; ++v)
        if (s_indeg[v][tid] == 0) s_order[tail++][tid] = (unsigned char)v;
    for (int head = 0; head < tail && head < N; ++head) {
        uint64_t c = s_child[s_order[head][tid]][tid];
        while (c) {
            const int v = dvs_ctz64(c);
            c &= c - 1;
            if (--s_indeg[v][tid] == 0 && tail < DVS_WTOK) s_order[tail++][tid] = (unsigned char)v;
        }
    }
    if (tail != N) bad |= 8;
    for (int v = 0; v < N; ++v) s_reach[v][tid] = s_child[v][tid] | (1ull << v);
    for (int k = 0; k < N; ++k) {
        const uint64_t rk = s_reach[k][tid];
        for (int v = 0; v < N; ++v)
            if ((s_reach[v][tid] >> k) & 1ull) s_reach[v][tid] |= rk;
    }
    DvsRecordW* r = a.rec + dag;
    for (int i = 0; i < DVS_WTOK; ++i) {
        int label = 0, pos = 0;
        uint64_t parents = 0, allowed = 1ull << i;
        if (i < N) {
            if (i == 0) label = 2;
            else if (i == 1) label = 0;
            else if (i == out_id) label = 1;
            else {
                label = a.labels[(size_t)dag * n + (i - 2)] + 3;
                if (label >= a.C) { bad |= 1; label = 0; }
            }
            pos = s_order[i][tid];
            allowed = 0;
            for (int j = 0; j < N; ++j) {
                if ((s_child[j][tid] >> i) & 1ull) parents |= 1ull << j;
                if ((s_reach[j][tid] >> i) & 1ull) allowed |= 1ull << j;
            }
        }
        r->label[i] = (uint8_t)label;
        r->pos[i] = (uint8_t)pos;
        r->parents[i] = parents;
        r->allowed[i] = allowed;
    }
    if (bad) atomicOr(a.status, bad);
}

void dvs_launch_build_records_w(const BuildWArgs& a, dvs_stream_t st) {
    DVS_LAUNCH(k_build_records_w, dim3((a.B + BW_T - 1) / BW_T), dim3(BW_T), 0, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Embedding forward (pace.py:201-221, 1181-1184, 1624-1630), wide.  Token-parallel: a wave owns one tile; the hidden
// of the positional encoder is an LDS GATHER over the token's parent bit-row:
//   e1[i] = relu(W1[pos_i] + sum_{j in parents(i)} W1[N + pos_j])
// (the one-tile kernel does the same with selector-matrix MFMAs; with up to 48 positions the gather is cheaper).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_embed_fwd_w(EmbedArgs a) {
    DVS_DYN_LDS(smem);
    const EmbWLds l = embw_lds(smem);
    embw_stage(l, a);
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N;
    const int ntiles = a.dims.B * a.dims.NT;
    for (int tile = blockIdx.x * L.nwaves + L.wave; tile < ntiles; tile += gridDim.x * L.nwaves) {
        const DvsTile T = dvs_tile_of(tile, a.dims);
        const DvsRecordW* rec = (const DvsRecordW*)a.rec + T.dag;
        const bool valid = L.r < T.Nl;
        f4 e1[4];
        embw_hidden(e1, l.W1, rec, N, T.tok0, T.Nl, L);
        const uint32_t gdag = a.dims.dag_offset + T.dag;
        dvs_dropout_tile(e1, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site, gdag), D, L, T.tok0);
        f4 x[4];
        f4 e2[2] = {f4_zero(), f4_zero()};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f4 w0 = dvs_wcol(l.W2, EMB_LDW2, 0, t, L), w1 = dvs_wcol(l.W2, EMB_LDW2, 16, t, L);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                e2[0] = dvs_mfma(w0[kk], e1[t][kk], e2[0]);
                e2[1] = dvs_mfma(w1[kk], e1[t][kk], e2[1]);
            }
        }
        {
            f4 tmp[4] = {e2[0], e2[1], f4_zero(), f4_zero()};
            dvs_dropout_tile(tmp, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site + 1, gdag), D, L, T.tok0);
            x[2] = tmp[0];
            x[3] = tmp[1];
        }
        const int label = rec->label[valid ? T.tok0 + L.r : 0];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int f = 16 * t + 4 * L.g + kk;
                x[t][kk] = valid ? fmaxf(l.labw[f * EMBW_LABLD + label] + l.labb[f], 0.f) : 0.f;
            }
        if (!valid) { x[2] = f4_zero(); x[3] = f4_zero(); }
        dvs_store_tile(a.out, tile, x, L);
    }
}

void dvs_launch_embed_fwd_w(const EmbedArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = EMBW_FLOATS * 4;
    DVS_SET_LDS(k_embed_fwd_w, lds);
    DVS_LAUNCH(k_embed_fwd_w, dim3(grid), dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Attention sublayer forward, wide (same math as k_attn_fwd: pace.py:52-56 / 144 / 148).
//   wave w < NT : x tile -> q, k, v (MFMA) -> parked in LDS      | barrier
//   all threads : one (token i, head h) item each: scores over the ancestor bit-row of i, softmax, dropout, O = P'V
//                 written to LDS                                  | barrier
//   wave w < NT : y = Wo O + bo, dropout, residual, LayerNorm statistics, store
// ---------------------------------------------------------------------------------------------------------
struct AttnWLds {
    float *Win, *Wout, *inb, *outb, *lg, *lb, *Q, *K, *V, *O;
    uint64_t* rows;              // [3][48]: ancestor bit-rows of the current DAG, their even / odd set bits
};
__device__ __forceinline__ AttnWLds attnw_lds(char* smem) {
    AttnWLds l;
    l.Win = (float*)smem;
    l.Wout = l.Win + 192 * DVS_LD;
    l.inb = l.Wout + 64 * DVS_LD;
    l.outb = l.inb + 192;
    l.lg = l.outb + 64;
    l.lb = l.lg + 64;
    l.Q = l.lb + 64;
    l.K = l.Q + DVS_WSCR;
    l.V = l.K + DVS_WSCR;
    l.O = l.V + DVS_WSCR;
    l.rows = (uint64_t*)(l.O + DVS_WSCR);        // float offset is even: 8-byte aligned
    return l;
}
constexpr size_t ATTNW_FLOATS = 256 * DVS_LD + 192 + 64 + 128 + 4 * (size_t)DVS_WSCR + 6 * DVS_WTOK;

// 8 waves: waves 0..NT-1 own the tiles (MFMA parts), all 8 share the (token, head) items of the core.
__global__ __launch_bounds__(512) void k_attn_fwd_w(AttnArgs a) {
    DVS_DYN_LDS(smem);
    const AttnWLds l = attnw_lds(smem);
    dvs_stage_matrix(l.Win, DVS_LD, a.in_w, 64, 192, 64);
    dvs_stage_matrix(l.Wout, DVS_LD, a.out_w, 64, 64, 64);
    dvs_stage_vector(l.inb, a.in_b, 192);
    dvs_stage_vector(l.outb, a.out_b, 64);
    if (a.ln.stats) {
        dvs_stage_vector(l.lg, a.ln.g, 64);
        dvs_stage_vector(l.lb, a.ln.b, 64);
    }
    for (int i = threadIdx.x; i < 4 * DVS_WSCR; i += blockDim.x) l.Q[i] = 0.f;
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N, NT = a.dims.NT, NTOK = 16 * NT;
    const int tok0 = 16 * L.wave, Nl = dvs_rows_of(N, L.wave);
    const bool has_tile = L.wave < NT;
    const float scale = 0.35355339059327373f;   // 1/sqrt(8)
    for (int dag = blockIdx.x; dag < a.dims.B; dag += gridDim.x) {
        const DvsRecordW* rec = (const DvsRecordW*)a.rec + dag;
        const size_t tile = (size_t)dag * NT + L.wave;
        const uint32_t gdag = a.dims.dag_offset + dag;
        if (threadIdx.x >= 256 && threadIdx.x < 256 + DVS_WTOK) {      // a wave without a tile prepares the rows
            const int i = threadIdx.x - 256;
            const uint64_t row = i < N ? rec->allowed[i] : 0ull;
            uint64_t e, o;
            dvs_split_row(row, e, o);
            l.rows[i] = row;
            l.rows[DVS_WTOK + i] = e;
            l.rows[2 * DVS_WTOK + i] = o;
        }
        f4 x[4];
        if (has_tile) {
            f4 kv[4], dummy[4];
            float rstd;
            dvs_load_x<false>(x, dummy, rstd, a.xin, a.ln, l.lg, l.lb, tile, Nl, L);
            if (a.kv) {
                dvs_load_tile(kv, a.kv, tile, L);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) kv[t] = x[t];
            }
            f4 q[4], k[4], v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                q[t] = dvs_vecT(l.inb, t, L);
                k[t] = dvs_vecT(l.inb + 64, t, L);
                v[t] = dvs_vecT(l.inb + 128, t, L);
            }
            dvs_mat_T<4, 4>(q, x, l.Win, DVS_LD, 0, L);
            dvs_mat_T<4, 4>(k, kv, l.Win, DVS_LD, 64, L);
            dvs_mat_T<4, 4>(v, kv, l.Win, DVS_LD, 128, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) q[t] *= scale;
            dvs_park_T(l.Q + tok0 * DVS_LD, q, L);
            dvs_park_T(l.K + tok0 * DVS_LD, k, L);
            dvs_park_T(l.V + tok0 * DVS_LD, v, L);
        }
        __syncthreads();
        const uint32_t kprob = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_prob, gdag);
        {
            const DvsCoreItem it = dvs_core_item(N);
            const bool active = it.tok >= 0;
            const int i = active ? it.tok : 0, h = it.head;
            const f4 q0 = *(const f4*)(l.Q + i * DVS_LD + 8 * h), q1 = *(const f4*)(l.Q + i * DVS_LD + 8 * h + 4);
            // ONE pass over the (half) ancestor row with an online softmax (running max m, denominator and output rescaled
            // when m grows), two keys per iteration so that their LDS reads overlap: the row walk is latency-bound.
            // Dropout acts on the normalised probabilities; it commutes with the final division by the denominator.
            float m = -3.0e38f, den = 0.f;
            f4 o0 = f4_zero(), o1 = f4_zero();
            for (uint64_t mm = active ? l.rows[(it.half + 1) * DVS_WTOK + i] : 0ull; mm;) {
                const int j0 = dvs_ctz64(mm);
                mm &= mm - 1;
                const bool two = mm != 0;
                const int j1 = two ? dvs_ctz64(mm) : j0;
                mm &= mm - 1;                                   // no-op on 0
                const float s0 = dvs_dot8(q0, q1, l.K + j0 * DVS_LD + 8 * h);
                const float s1 = two ? dvs_dot8(q0, q1, l.K + j1 * DVS_LD + 8 * h) : -3.0e38f;
                const float mn = fmaxf(m, fmaxf(s0, s1));
                const float sc = __expf(m - mn);
                float e0 = __expf(s0 - mn), e1 = two ? __expf(s1 - mn) : 0.f;
                den = den * sc + (e0 + e1);
                if (D.on) {
                    e0 = dvs_dropout_elem(e0, kprob, (uint32_t)((h * NTOK + i) * NTOK + j0), D);
                    e1 = dvs_dropout_elem(e1, kprob, (uint32_t)((h * NTOK + i) * NTOK + j1), D);
                }
                const float* v0 = l.V + j0 * DVS_LD + 8 * h;
                const float* v1 = l.V + j1 * DVS_LD + 8 * h;
                o0 = o0 * sc + *(const f4*)v0 * e0 + *(const f4*)v1 * e1;
                o1 = o1 * sc + *(const f4*)(v0 + 4) * e0 + *(const f4*)(v1 + 4) * e1;
                m = mn;
            }
            // merge the two halves of a split row (all lanes execute the exchange; whole-row lanes ignore it)
            const float pm = dvs_pair_xchg(m), pden = dvs_pair_xchg(den);
            const f4 po0 = dvs_pair_xchg(o0), po1 = dvs_pair_xchg(o1);
            if (it.half >= 0) {
                const float M = fmaxf(m, pm);
                const float fa = __expf(m - M), fb = __expf(pm - M);
                den = den * fa + pden * fb;
                o0 = o0 * fa + po0 * fb;
                o1 = o1 * fa + po1 * fb;
            }
            if (active && it.half <= 0) {
                const float rden = 1.0f / den;
                *(f4*)(l.O + i * DVS_LD + 8 * h) = o0 * rden;
                *(f4*)(l.O + i * DVS_LD + 8 * h + 4) = o1 * rden;
            }
        }
        __syncthreads();
        if (has_tile) {
            f4 o[4], y[4];
            dvs_lds_T(o, l.O, tok0, L);
#pragma unroll
            for (int t = 0; t < 4; ++t) y[t] = dvs_vecT(l.outb, t, L);
            dvs_mat_T<4, 4>(y, o, l.Wout, DVS_LD, 0, L);
            dvs_dropout_tile(y, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L, tok0);
            const bool valid = L.r < Nl;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) y[t][kk] = valid ? x[t][kk] + y[t][kk] : 0.f;
            dvs_store_pre(a.out_pre, a.out_stats, tile, y, L);
        }
    }
}

void dvs_launch_attn_fwd_w(const AttnArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = ATTNW_FLOATS * 4;
    DVS_SET_LDS(k_attn_fwd_w, lds);
    DVS_LAUNCH(k_attn_fwd_w, dim3(grid), dim3(512), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Loss head forward, wide (pace.py:1880-1972; k_loss_fwd restated for 3 tiles, up to 48 classes).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_loss_fwd_w(LossArgs a) {
    DVS_DYN_LDS(smem);
    const LossWLds l = lossw_lds(smem);
    lossw_stage(l, a);
    __syncthreads();
    const Lane L = dvs_lane();
    const int N = a.dims.N, C = a.dims.C, NT = a.dims.NT;
    const int tok0 = 16 * L.wave, Nl = dvs_rows_of(N, L.wave);
    const bool has_tile = L.wave < NT;
    const float b2 = l.b2[0];
    for (int dag = blockIdx.x; dag < a.dims.B; dag += gridDim.x) {
        const DvsRecordW* rec = (const DvsRecordW*)a.rec + dag;
        const size_t tile = (size_t)dag * NT + L.wave;
        float nll = 0.f;
        f4 U[4], w2v[4];
        const int tok = tok0 + L.r;                       // this lane's token
        if (has_tile) {
            f4 h[4], dummy[4];
            float rstd;
            dvs_load_x<false>(h, dummy, rstd, a.xin, a.ln, l.lg, l.lb, tile, Nl, L);
            // ---- node head: 3 class tiles ------------------------------------------------------------------------
            f4 t1[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) t1[t] = dvs_vecT(l.bn1, t, L);
            dvs_mat_T<2, 4>(t1, h, l.Wn1, DVS_LD, 0, L);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) t1[t][kk] = fmaxf(t1[t][kk], 0.f);
            f4 lgt[3];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct) lgt[ct] = *(const f4*)(l.bn2 + 16 * ct + 4 * L.g);
            dvs_mat_T<3, 2>(lgt, t1, l.Wn2, LOSSW_LDN2, 0, L);
            float mx = -3.0e38f;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) mx = (16 * ct + 4 * L.g + reg < C) ? fmaxf(mx, lgt[ct][reg]) : mx;
            mx = dvs_max_g(mx);
            float se = 0.f;
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) se += (16 * ct + 4 * L.g + reg < C) ? __expf(lgt[ct][reg] - mx) : 0.f;
            se = dvs_sum_g(se);
            const float lse = mx + __logf(se);
            const int target = rec->label[tok + 1 < DVS_WTOK ? tok + 1 : 0];
#pragma unroll
            for (int ct = 0; ct < 3; ++ct)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    nll -= (16 * ct + 4 * L.g + reg == target && tok < N - 1) ? (lgt[ct][reg] - lse) : 0.f;
            // ---- edge head: U = Wa h (registers), V = Wb h + b1 (LDS, all tiles) -----------------------------------
            f4 V[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                U[t] = f4_zero();
                V[t] = dvs_vecT(l.be1, t, L);
                w2v[t] = dvs_vecT(l.w2, t, L);
            }
            dvs_mat_T<4, 4>(U, h, l.Wa, DVS_LD, 0, L);
            dvs_mat_T<4, 4>(V, h, l.Wb, DVS_LD, 0, L);
            dvs_park_T(l.V + tok0 * DVS_LD, V, L);
        }
        __syncthreads();
        if (has_tile) {
            const uint64_t par = rec->parents[tok + 1 < DVS_WTOK ? tok + 1 : 0];
            float enll = 0.f;
            const int jend = (N - 2 < tok0 + 15) ? N - 2 : tok0 + 15;     // pairs need j < i <= tok0 + 15
            for (int j = 0; j < jend; ++j) {
                float e = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f4 vj = *(const f4*)(l.V + j * DVS_LD + 16 * t + 4 * L.g);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) e += w2v[t][kk] * fmaxf(U[t][kk] + vj[kk], 0.f);
                }
                const float logit = dvs_sum_g(e) + b2;
                const bool pv = (tok > j) && (tok <= N - 2);
                const float truth = (float)((par >> (j + 1)) & 1ull);
                const float bce = fmaxf(logit, 0.f) - logit * truth + log1pf(__expf(-fabsf(logit)));
                enll += pv ? bce : 0.f;
            }
            nll += (L.g == 0) ? enll : 0.f;
            nll = dvs_sum_wave(nll);
            if (L.lane == 0) l.part[L.wave] = nll;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            float s = 0.f;
            for (int w = 0; w < NT; ++w) s += l.part[w];
            a.dag_loss[(size_t)dag * 2] = s;
        }
    }
}

void dvs_launch_loss_fwd_w(const LossArgs& a, int grid, dvs_stream_t st) {
    const size_t lds = dvs_lossw_lds_floats() * 4;
    DVS_SET_LDS(k_loss_fwd_w, lds);
    DVS_LAUNCH(k_loss_fwd_w, dim3(grid), dim3(256), lds, st, a);
}
