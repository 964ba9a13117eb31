// Forward kernels of the PACE-VAE step: feature packing, embeddings, attention sublayer, FFN sublayer.
// One wave owns one DAG (see dvs_device.h); workgroups are persistent and keep the sublayer's weights in LDS.
#include "dvs_kernels.h"
#include "dvs_wimg.h"

// ---------------------------------------------------------------------------------------------------------
// dvs_pack_features: reference-layout dense features -> 96-byte records (replaces pace.py:1981-1985's
// .to(device) hand-over; the features themselves come from prepare_features, pace.py:1345-1478).
// One thread per (mask head, DAG, token slot).
// ---------------------------------------------------------------------------------------------------------
// A workgroup owns PACK_DAGS consecutive DAGs: their label / position / adjacency rows and their 8 per-head mask
// copies are four CONTIGUOUS byte ranges of the batched feature tensors, streamed into LDS with 16-byte loads
// (fully coalesced; the HBM-bound leg of the step: 4.5 KB per DAG at n=12), then one thread per (DAG, token) builds
// its record fields from LDS.
constexpr int PACK_DAGS = 4;     // 18 KB of LDS, 512 threads = (8 mask heads) x (4 DAGs) x (16 tokens)
__device__ __forceinline__ void pack_stream(float* dst, const float* __restrict__ src, size_t nfloats) {
    const size_t n4 = nfloats >> 2;
    size_t i = threadIdx.x;
    for (; i + 3 * blockDim.x < n4; i += 4 * blockDim.x) {      // four 16-byte loads in flight per lane
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f4*)(src + 4 * (i + u * blockDim.x));
#pragma unroll
        for (int u = 0; u < 4; ++u) *(f4*)(dst + 4 * (i + u * blockDim.x)) = v[u];
    }
    for (; i < n4; i += blockDim.x) *(f4*)(dst + 4 * i) = *(const f4*)(src + 4 * i);
    for (size_t i = 4 * n4 + threadIdx.x; i < nfloats; i += blockDim.x) dst[i] = src[i];
}
__global__ __launch_bounds__(8 * 16 * PACK_DAGS) void k_pack(PackArgs a) {
    DVS_DYN_LDS(smem);
    const int N = a.N, C = a.C;
    const int dag0 = blockIdx.x * PACK_DAGS;
    const int nd = (a.B - dag0 < PACK_DAGS) ? a.B - dag0 : PACK_DAGS;
    float* s_lab = (float*)smem;                       // [PACK_DAGS][N][C]
    float* s_pos = s_lab + PACK_DAGS * 16 * 16;        // [PACK_DAGS][N][N]
    float* s_adj = s_pos + PACK_DAGS * 16 * 16;
    float* s_msk = s_adj + PACK_DAGS * 16 * 16;        // [PACK_DAGS][8][N][N] bytes, streamed as floats
    // DAG dag0's blocks start at multiples of PACK_DAGS = 4 DAGs * {N*C, N*N} floats (8*N*N mask bytes): multiples of 16 bytes
    // the four ranges in ONE batch of loads (four pack_stream calls in a row were four memory round trips for ~2 loads per lane)
    {
        const float* src[4] = {a.lab1h + (size_t)dag0 * N * C, a.pos1h + (size_t)dag0 * N * N, a.adj + (size_t)dag0 * N * N,
                               (const float*)(a.tmask + (size_t)dag0 * 8 * N * N)};
        float* dst[4] = {s_lab, s_pos, s_adj, s_msk};
        const int nf[4] = {nd * N * C, nd * N * N, nd * N * N, nd * 8 * N * N / 4};
        bool whole = true;                                        // every range a whole number of 16-byte pieces, <= 2 per lane
#pragma unroll
        for (int k = 0; k < 4; ++k) whole = whole && (nf[k] & 3) == 0 && (nf[k] >> 2) <= 2 * (int)blockDim.x;
        if (whole) {
            f4 v[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = (int)threadIdx.x + u * (int)blockDim.x;
                    v[k][u] = *(const f4*)(src[k] + 4 * (i < (nf[k] >> 2) ? i : 0));
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int i = (int)threadIdx.x + u * (int)blockDim.x;
                    if (i < (nf[k] >> 2)) *(f4*)(dst[k] + 4 * i) = v[k][u];
                }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) pack_stream(dst[k], src[k], (size_t)nf[k]);
        }
    }
    __syncthreads();
    // thread = (head h, DAG d, token i): head 0 builds the record row, heads 1..7 check that their mask copy equals head 0's
    const int h = threadIdx.x / (16 * PACK_DAGS), d = (threadIdx.x >> 4) % PACK_DAGS, i = threadIdx.x & 15;
    if (d >= nd) return;
    const uint8_t* msk = (const uint8_t*)s_msk;
    int bad = 0;
    if (h > 0) {
        if (i < N) {
            const uint8_t* m0 = msk + ((size_t)d * 8 * N + i) * N;
            const uint8_t* mh = msk + (((size_t)d * 8 + h) * N + i) * N;
            for (int j = 0; j < N; ++j)
                if ((mh[j] != 0) != (m0[j] != 0)) bad |= 2;
        }
        if (bad) atomicOr(a.status, bad);
        return;
    }
    int label = 0, pos = 0;
    unsigned parents = 0, allowed = 1u << i;
    if (i < N) {
        const float* lr = s_lab + ((size_t)d * N + i) * C;
        int ones = 0;
        for (int c = 0; c < C; ++c) {
            const float v = lr[c];
            if (v == 1.0f) { label = c; ++ones; } else if (v != 0.0f) bad |= 1;
        }
        if (ones != 1) bad |= 1;
        const float* pr = s_pos + ((size_t)d * N + i) * N;
        ones = 0;
        for (int c = 0; c < N; ++c) {
            const float v = pr[c];
            if (v == 1.0f) { pos = c; ++ones; } else if (v != 0.0f) bad |= 1;
        }
        if (ones != 1) bad |= 1;
        const float* ad = s_adj + (size_t)d * N * N;
        for (int j = 0; j < N; ++j)
            if (ad[j * N + i] != 0.0f) parents |= 1u << j;
        allowed = 0;
        const uint8_t* m0 = msk + ((size_t)d * 8 * N + i) * N;
        for (int j = 0; j < N; ++j)
            if (!m0[j]) allowed |= 1u << j;
        if (!((allowed >> i) & 1u)) bad |= 4;
    }
    DvsRecord* r = a.rec + dag0 + d;
    r->label[i] = (uint8_t)label;
    r->pos[i] = (uint8_t)pos;
    r->parents[i] = (uint16_t)parents;
    r->allowed[i] = (uint16_t)allowed;
    if (bad) atomicOr(a.status, bad);
}

void dvs_launch_pack(const PackArgs& a, dvs_stream_t st) {
    const size_t lds = (size_t)PACK_DAGS * (3 * 256 * 4 + 8 * 256);
    DVS_SET_LDS(k_pack, lds);
    DVS_LAUNCH(k_pack, dim3((a.B + PACK_DAGS - 1) / PACK_DAGS), dim3(8 * 16 * PACK_DAGS), lds, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// dvs_build_records: row codec -> records on the device, one thread per DAG.  Restates, for N <= 16
// tokens held as 16-bit rows:
//   PACE wrapping   (pace.py:1250-1288): v0 = start(2), v1 = input(0), v_{N-1} = output(1), user k -> k+2, label+3;
//                   sources hang off v1, vertices without children feed v_{N-1}
//   positions       (pace.py:1245-1248, 1286): FIFO Kahn order (zero in-degree vertices in id order, children relaxed in
//                   ascending id) and the quirk positions[v] = order[v]
//   ancestor mask   (pace.py:1307-1343 + .transpose at 1474): token i may attend j iff j reaches i or j == i
// Per-thread arrays live in LDS ([entry][thread] so that a wave's accesses are conflict-free).
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_build_records(BuildArgs a) {
    __shared__ unsigned short s_child[16][256];
    __shared__ unsigned short s_reach[16][256];
    __shared__ unsigned char s_indeg[16][256];
    __shared__ unsigned char s_order[16][256];
    const int tid = threadIdx.x;
    const int dag = blockIdx.x * 256 + tid;
    if (dag >= a.B) return;
    const int N = a.N, n = N - 3, out_id = N - 1;
    int bad = 0;
    for (int v = 0; v < 16; ++v) s_child[v][tid] = 0;
    s_child[0][tid] = 1u << 1;
    unsigned haspred = 0;
    for (int v = 0; v < n; ++v) {
        const unsigned p = a.preds[(size_t)dag * n + v];
        if (p >> v) bad |= 8;                                   // an edge u -> v needs u < v
        for (int u = 0; u < v; ++u)
            if ((p >> u) & 1u) s_child[u + 2][tid] |= (unsigned short)(1u << (v + 2));
        if (p & ((1u << v) - 1u)) haspred |= 1u << v;
    }
    for (int v = 0; v < n; ++v)
        if (!((haspred >> v) & 1u)) s_child[1][tid] |= (unsigned short)(1u << (v + 2));
    for (int v = 0; v < N - 1; ++v)
        if (s_child[v][tid] == 0) s_child[v][tid] = (unsigned short)(1u << out_id);
    // in-degrees and FIFO Kahn
    for (int v = 0; v < 16; ++v) s_indeg[v][tid] = 0;
    for (int u = 0; u < N; ++u) {
        unsigned c = s_child[u][tid];
        while (c) {
            const int v = __ffs((int)c) - 1;
            c &= c - 1;
            s_indeg[v][tid]++;
        }
    }
    int tail = 0;
    for (int v = 0; v < N; ++v)
        if (s_indeg[v][tid] == 0) s_order[tail++][tid] = (unsigned char)v;
    for (int head = 0; head < tail && head < N; ++head) {
        unsigned c = s_child[s_order[head][tid]][tid];
        while (c) {
            const int v = __ffs((int)c) - 1;
            c &= c - 1;
            if (--s_indeg[v][tid] == 0 && tail < 16) s_order[tail++][tid] = (unsigned char)v;
        }
    }
    if (tail != N) bad |= 8;
    // reach[a] = descendants-or-self of a (Warshall on bit rows)
    for (int v = 0; v < N; ++v) s_reach[v][tid] = (unsigned short)(s_child[v][tid] | (1u << v));
    for (int k = 0; k < N; ++k) {
        const unsigned rk = s_reach[k][tid];
        for (int v = 0; v < N; ++v)
            if ((s_reach[v][tid] >> k) & 1u) s_reach[v][tid] |= (unsigned short)rk;
    }
    DvsRecord* r = a.rec + dag;
    for (int i = 0; i < 16; ++i) {
        int label = 0, pos = 0;
        unsigned parents = 0, allowed = 1u << i;
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
                if ((s_child[j][tid] >> i) & 1u) parents |= 1u << j;
                if ((s_reach[j][tid] >> i) & 1u) allowed |= 1u << j;
            }
        }
        r->label[i] = (uint8_t)label;
        r->pos[i] = (uint8_t)pos;
        r->parents[i] = (uint16_t)parents;
        r->allowed[i] = (uint16_t)allowed;
    }
    if (bad) atomicOr(a.status, bad);
}

void dvs_launch_build_records(const BuildArgs& a, dvs_stream_t st) {
    DVS_LAUNCH(k_build_records, dim3((a.B + 255) / 256), dim3(256), 0, st, a);
}

// ---------------------------------------------------------------------------------------------------------
// Embedding (GnnPositionalEncoding.forward pace.py:201-221 + vertex_label_embed 1181-1184 + cat 1624-1630).
// One-hot inputs make both first layers row gathers:
//   e1[i] = relu(W1[pos_i] + sum_{j parent of i} W1[N + pos_j]);  e2 = drop(drop(e1) @ W2)
//   le[i] = relu(lab_w[:, label_i] + lab_b);                      x0 = cat(le, e2)
// ---------------------------------------------------------------------------------------------------------
struct EmbLds {
    float *W1, *W2, *labw, *labb, *scr;
};
__device__ __forceinline__ EmbLds emb_lds(char* smem) {
    EmbLds l;
    l.W1 = (float*)smem;
    l.W2 = l.W1 + 2 * DVS_MAXTOK * DVS_LD;
    l.labw = l.W2 + 64 * EMB_LDW2;
    l.labb = l.labw + 32 * 16;
    l.scr = l.labb + 32;
    return l;
}
static size_t emb_lds_floats(int nwaves) {          // at least the padded block (DvsEmbImg::FLOATS): the copy is verbatim
    const size_t n = 2 * DVS_MAXTOK * DVS_LD + 64 * EMB_LDW2 + 32 * 16 + 32 + (size_t)nwaves * 16;
    return n > (size_t)DvsEmbImg::FLOATS ? n : (size_t)DvsEmbImg::FLOATS;
}

__global__ __launch_bounds__(1024) void k_embed_fwd(EmbedArgs a, DvsStagePlan plan) {
    DVS_DYN_LDS(smem);
    const int N = a.dims.N, C = a.dims.C;
    const EmbLds l = emb_lds(smem);
    // the embedding block (dvs_wimg.h) in one batch: 20 wave chunks on 16 waves (4 in the narrow mapping of small batches)
    if (blockDim.x >= 1024) dvs_stage_now<2>(&plan, smem);
    else dvs_stage_now<5>(&plan, smem);
    __syncthreads();
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    float* scr = l.scr + L.wave * 16;
    for (int dag = blockIdx.x * L.nwaves + L.wave; dag < a.dims.B; dag += gridDim.x * L.nwaves) {
        const DvsRecord* rec = a.rec + dag;
        const bool valid = L.r < N;
        const EmbSel sel = dvs_emb_selectors(rec, N, scr, L);
        f4 eh[4];
        dvs_emb_hidden(eh, l.W1, N, sel, L);
        const uint32_t gdag = a.dims.dag_offset + dag;
        const int label = rec->label[L.r];
        // selectors and hidden layer are shared; the encoder-side and (train mode) decoder-side embeddings differ only in
        // their dropout sites
        for (int v = 0; v < (a.out2 ? 2 : 1); ++v) {
            const int site = v ? a.site2 : a.site;
            f4 e1[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) e1[t] = eh[t];
            dvs_dropout_tile(e1, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, site, gdag), D, L);
            f4 x[4];
            // positional half: e2^T[32 x tok] = W2^T e1^T  (A = W2 column fragments)
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
                f4 tmp[2] = {e2[0], e2[1]};
                dvs_dropout_tile<2>(tmp, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, site + 1, gdag), D, L);
                x[2] = tmp[0];
                x[3] = tmp[1];
            }
            // label half
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int f = 16 * t + 4 * L.g + kk;
                    x[t][kk] = valid ? fmaxf(l.labw[f * 16 + label] + l.labb[f], 0.f) : 0.f;
                }
            if (!valid) { x[2] = f4_zero(); x[3] = f4_zero(); }
            dvs_store_tile(v ? a.out2 : a.out, dag, x, L);
        }
    }
}

void dvs_launch_embed_fwd(const EmbedArgs& a, int grid, int nw, dvs_stream_t st) {      // nw: 16 waves per workgroup, or 4
    const size_t lds = emb_lds_floats(16) * 4;
    DvsStagePlan plan;
    dvs_plan_clear(plan);
    dvs_plan_seg(plan, DVS_FAKE_LDS, DVS_FAKE_LDS, a.embimg, 2 * DvsEmbImg::FLOATS);
    dvs_plan_seal(plan);
    DVS_SET_LDS(k_embed_fwd, lds);
    DVS_LAUNCH(k_embed_fwd, dim3(grid), dim3(nw == 4 ? 256 : 1024), lds, st, a, plan);
}

// ---------------------------------------------------------------------------------------------------------
// Attention sublayer forward: x = LN_prev(pre_prev); y = MHA(x, kv, kv, mask) (nn.MultiheadAttention explicit
// path, pace.py:52-56 / 144 / 148); pre = x + dropout(y); stats(pre).  Everything between the tile load and the
// tile store stays in registers:  q^T,k^T (T) -> S^T = K Q^T per head (lane r = query i, regs = keys 4g..4g+3)
// -> softmax over keys (in-lane + 2 shuffles) -> O^T = V^T P^T -> y^T = Wo O^T.
// ---------------------------------------------------------------------------------------------------------
#ifdef DVS_STAMPS
DVS_STAMP_DECL(dvs_stamps_fwd);
struct DvsLatTag { int phase; };
__device__ const DvsLatTag dvs_lat_tag = {6};
#define DVS_LAT_STAMP(id) DVS_STAMP(dvs_stamps_fwd, &dvs_lat_tag, id)
// inner budget of the attention forward DAG loop: cycles summed over DAGs and phases, per (workgroup, wave, segment)
__device__ unsigned long long dvs_stamps_attn[256 * 8 * 8];
#define ASTAMP(k)                                                                                                   \
    do {                                                                                                            \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                               \
        if ((dvs_tid() & 63) == 0 && dvs_bid() < 256) dvs_stamps_attn[(dvs_bid() * 8 + (dvs_tid() >> 6)) * 8 + (k)] += now_ - ast_; \
        ast_ = now_;                                                                                                \
    } while (0)
#else
#define ASTAMP(k) ((void)0)
#endif
#include "dvs_latent.h"
struct AttnLds {
    dvs_bf16 *Win, *Wout;                     // bf16x6 image triples (dvs_bf16.h); in-projection rows / out-projection columns in slot order
    float *inb, *outb, *lg, *lb;
};
DVS_HD inline AttnLds attn_lds(char* smem) {
    AttnLds l;
    l.Win = (dvs_bf16*)smem;
    l.Wout = l.Win + 3 * 192 * DVS_LDB;
    l.inb = (float*)(l.Wout + 3 * 64 * DVS_LDB);
    l.outb = l.inb + 192;
    l.lg = l.outb + 64;
    l.lb = l.lg + 64;
    return l;
}
static size_t attn_lds_bytes() { return 3 * 256 * DVS_LDB * sizeof(dvs_bf16) + (192 + 64 + 128) * sizeof(float); }

// staging plan (dvs_stage.h), built on the host by the launchers
inline void attn_plan(DvsStagePlan& p, const AttnArgs& a, char* smem) {
    const AttnLds l = attn_lds(smem);
    dvs_plan_clear(p);
    // Win (rows in head-aligned slot order) and Wout (columns likewise) as ready-made bf16x6 images: one straight copy
    dvs_plan_seg(p, smem, l.Win, (const dvs_bf16*)a.wimg + DvsAttnImg::Win, (int)(DvsAttnImg::WoutT - DvsAttnImg::Win));
    dvs_plan_vec(p, smem, l.inb, a.in_b, 192, true);
    dvs_plan_vec(p, smem, l.outb, a.out_b, 64);
    dvs_plan_vec(p, smem, l.lg, a.ln.g, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.lb, a.ln.b, a.ln.stats ? 64 : 0);
    dvs_plan_seal(p);
}

// q^T, k^T (T-layout, q pre-scaled by 1/sqrt(dh)) and v (N-layout) of one DAG; fp32-accurate bf16x6 products
__device__ __forceinline__ void attn_qkv(f4 (&q)[4], f4 (&k)[4], f4 (&v)[4], const Split3T& x, const Split3T& kv,
                                         const AttnLds& l, const Lane& L) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        q[t] = dvs_vecT(l.inb, t, L);
        k[t] = dvs_vecT(l.inb + 64, t, L);
        v[t] = f4_splat(l.inb[128 + 16 * t + L.r]);
    }
    dvs_matb3<4>(q, x, l.Win, 192, 0, L);
    dvs_matb3<4>(k, kv, l.Win, 192, 64, L);
    dvs_matb3<4, true>(v, kv, l.Win, 192, 128, L);
    const float scale = 0.35355339059327373f;   // 1/sqrt(8)
#pragma unroll
    for (int t = 0; t < 4; ++t) q[t] *= scale;
}

// Attention core, all 8 heads, "transposed" orientation (lane r = query i, register reg = key 4g+reg).
// Phase-structured so that independent work is adjacent for the scheduler: 8 score chains, then 8 softmaxes, then
// 4 output-tile chains.  p[h][reg] = softmax_j(S_h[i=r][j=4g+reg]) before dropout; m/den = row max / denominator.
__device__ __forceinline__ void attn_scores_T(f4 (&s)[8], const f4 (&q)[4], const f4 (&k)[4], const Lane& L) {
#pragma unroll
    for (int h = 0; h < 8; ++h) s[h] = f4_zero();
    // slot order: step kk of tile t contracts 4 features of head 2t + (kk >> 1)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            s[2 * t] = dvs_mfma(k[t][kk], q[t][kk], s[2 * t]);
            s[2 * t + 1] = dvs_mfma(k[t][kk + 2], q[t][kk + 2], s[2 * t + 1]);
        }
}
__device__ __forceinline__ void attn_softmax_T(f4 (&p)[8], float (&m)[8], float (&den)[8], const f4 (&s)[8],
                                               unsigned allowed_r, const Lane& L) {
    bool ok[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ok[reg] = (allowed_r >> (4 * L.g + reg)) & 1u;
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        float mx = -3.0e38f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) mx = ok[reg] ? fmaxf(mx, s[h][reg]) : mx;
        m[h] = mx;
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) m[h] = dvs_max_x16(m[h]);
#pragma unroll
    for (int h = 0; h < 8; ++h) m[h] = dvs_max_x32(m[h]);
#pragma unroll
    for (int h = 0; h < 8; ++h) {
        float sum = 0.f;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            p[h][reg] = ok[reg] ? __expf(s[h][reg] - m[h]) : 0.f;
            sum += p[h][reg];
        }
        den[h] = sum;
    }
#pragma unroll
    for (int h = 0; h < 8; ++h) den[h] = dvs_add_x16(den[h]);
#pragma unroll
    for (int h = 0; h < 8; ++h) den[h] = dvs_add_x32(den[h]);
#pragma unroll
    for (int h = 0; h < 8; ++h) p[h] *= (1.0f / den[h]);
}

// dropout on the probabilities of head h in the transposed orientation: element ((h*16 + i)*16 + j)
__device__ __forceinline__ f4 attn_drop_T(f4 p, uint32_t key, int h, const DvsDrop& D, const Lane& L) {
    if (!D.on) return p;
    const uint32_t p0 = (uint32_t)((h * 16 + L.r) * 8 + 2 * L.g);
    const uint32_t h0 = dvs_draw(key, p0), h1 = dvs_draw(key, p0 + 1);
    p[0] = ((h0 & 0xFFFFu) >= D.thr16) ? p[0] * D.scale : 0.f;
    p[1] = ((h0 >> 16) >= D.thr16) ? p[1] * D.scale : 0.f;
    p[2] = ((h1 & 0xFFFFu) >= D.thr16) ? p[2] * D.scale : 0.f;
    p[3] = ((h1 >> 16) >= D.thr16) ? p[3] * D.scale : 0.f;
    return p;
}

#ifndef DVS_ATTN_FWD_THREADS
#define DVS_ATTN_FWD_THREADS 512
#endif
// mine / stage_mine, next / has_next: staging plans of this phase and of the one that follows in a chained launch
// (dvs_stage.h): the next phase's images are fetched into registers behind this phase's DAG loop, ahead of the barrier.
// NW: waves per workgroup the phase is compiled for (8, or 4 in the narrow mapping of small batches, dvs_api.hip): sizes the
// first phase's staging batch; the DAG loop itself reads the workgroup width at run time (L.nwaves)
template <int NW, class PP>
__device__ __forceinline__ void dvs_attn_fwd_phase(const AttnArgs& a, char* smem, PP mine, bool stage_mine, PP next,
                                                   bool has_next) {
    const AttnLds l = attn_lds(smem);
    DVS_STAMP(dvs_stamps_fwd, mine, 0);
    if (stage_mine) {
        dvs_stage_now<(NW >= 8 ? DVS_PF_FWD : DVS_PF_FWD_TAIL)>(mine, smem);
        __syncthreads();
    }
    DVS_STAMP(dvs_stamps_fwd, mine, 1);
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int N = a.dims.N;
#ifdef DVS_STAMPS
    unsigned long long ast_ = __builtin_amdgcn_s_memtime();
#endif
    bool gate = !stage_mine;                 // chained: the barrier that publishes this phase's images (DVS_PHASE_GATE)
    DVS_PHASE_GATE_INIT(gate);
    dvs_stagger(L.wave);
    // every wave of the workgroup runs the same number of rounds (the gate is a workgroup barrier): a wave beyond the batch
    // in the last round skips its DAG inside the round
    for (int base = dvs_bid() * L.nwaves; base < a.dims.B; base += gridDim.x * L.nwaves) {
        const int dag = base + L.wave;
        if (dag >= a.dims.B) {
            DVS_PHASE_GATE(gate);
            continue;
        }
        ASTAMP(7);
        f4 x[4], dummy[4], kvt[4];
        float rstd;
        {
            DvsRawX rx;
            dvs_load_x_issue(rx, a.xin, a.ln, dag, L);
            if (a.kv) dvs_load_tile(kvt, a.kv, dag, L);
            DVS_PHASE_GATE(gate);              // behind the round's global loads, ahead of the first LDS access
            dvs_load_x_finish<false>(x, dummy, rstd, rx, a.ln, l.lg, l.lb, N, L);
        }
#ifdef DVS_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        ASTAMP(0);
        f4 q[4], k[4], v[4];
        {
            const Split3T xs = dvs_split3_T(x);
            if (a.kv) {
                attn_qkv(q, k, v, xs, dvs_split3_T(kvt), l, L);
            } else {
                attn_qkv(q, k, v, xs, xs, l, L);
            }
        }
        ASTAMP(1);
        const unsigned allowed_r = a.rec[dag].allowed[L.r];
        const uint32_t gdag = a.dims.dag_offset + dag;
        const uint32_t kprob = dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_prob, gdag);
        f4 s[8], p[8];
        float m[8], den[8];
        attn_scores_T(s, q, k, L);
        attn_softmax_T(p, m, den, s, allowed_r, L);
        ASTAMP(2);
#pragma unroll
        for (int h = 0; h < 8; ++h) p[h] = attn_drop_T(p[h], kprob, h, D, L);
        ASTAMP(3);
        // O^T = V^T P^T per head on all 16 feature rows of the tile; rows reg 0,1 belong to head 2t, rows 2,3 to 2t+1
        f4 o[4];
        {
            f4 oa[4], ob[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) oa[t] = ob[t] = f4_zero();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    oa[t] = dvs_mfma(v[t][kk], p[2 * t][kk], oa[t]);
                    ob[t] = dvs_mfma(v[t][kk], p[2 * t + 1][kk], ob[t]);
                }
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t] = f4{oa[t][0], oa[t][1], ob[t][2], ob[t][3]};
        }
        ASTAMP(4);
        f4 y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = dvs_vecT(l.outb, t, L);
        dvs_matb3<4>(y, dvs_split3_T(o), l.Wout, 64, 0, L);
        ASTAMP(5);
        dvs_dropout_tile(y, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L);
        const bool valid = L.r < N;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) y[t][kk] = valid ? x[t][kk] + y[t][kk] : 0.f;
        dvs_store_pre(a.out_pre, a.out_stats, dag, y, L);
        ASTAMP(6);
    }
    DVS_PHASE_GATE(gate);                    // a workgroup without a DAG
    DVS_STAMP(dvs_stamps_fwd, mine, 2);
    if (has_next) {
        // the older wave group fetches the next phase's images while it waits for the younger one (dvs_stage.h)
        DvsPrefetch<DVS_PF_FWD_TAIL> pf;
        const bool fetcher = dvs_tid() < DVS_PF_THREADS;
        if (fetcher) dvs_prefetch_issue<false>(pf, next, dvs_tid(), DVS_PF_THREADS);
        DVS_STAMP(dvs_stamps_fwd, mine, 3);
        dvs_lds_barrier();               // every wave is done with this phase's images
        DVS_STAMP(dvs_stamps_fwd, mine, 4);
#ifdef DVS_STAMPS_ICACHE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DVS_STAMP(dvs_stamps_fwd, mine, 5);
#endif
        if (fetcher) dvs_prefetch_commit<false>(pf, next, smem, dvs_tid(), DVS_PF_THREADS);
        DVS_STAMP(dvs_stamps_fwd, mine, 6);
#ifdef DVS_STAMPS_ICACHE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (fetcher) dvs_prefetch_commit<false>(pf, next, smem, dvs_tid(), DVS_PF_THREADS);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DVS_STAMP(dvs_stamps_fwd, mine, 7);
#endif
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void k_attn_fwd(AttnArgs a, DvsStagePlan plan) {
    DVS_DYN_LDS(smem);
    dvs_attn_fwd_phase<NW>(a, smem, &plan, true, &plan, false);
}

int dvs_attn_fwd_waves() { return DVS_ATTN_FWD_THREADS / 64; }
void dvs_launch_attn_fwd(const AttnArgs& a, int grid, int nw, dvs_stream_t st) {
    const size_t lds = attn_lds_bytes();
    DvsStagePlan plan;
    attn_plan(plan, a, DVS_FAKE_LDS);
    if (nw == 4) {
        DVS_SET_LDS(k_attn_fwd<4>, lds);
        DVS_LAUNCH_AS("k_attn_fwd", k_attn_fwd<4>, dim3(grid), dim3(256), lds, st, a, plan);
    } else {
        DVS_SET_LDS(k_attn_fwd<DVS_ATTN_FWD_THREADS / 64>, lds);
        DVS_LAUNCH_AS("k_attn_fwd", k_attn_fwd<DVS_ATTN_FWD_THREADS / 64>, dim3(grid), dim3(DVS_ATTN_FWD_THREADS), lds, st, a, plan);
    }
}

// ---------------------------------------------------------------------------------------------------------
// FFN sublayer forward (pace.py:62-65 / 151-153): pre = x + drop(W2 drop(relu(W1 x + b1)) + b2)
// ---------------------------------------------------------------------------------------------------------
struct FfnLds {
    dvs_bf16 *W1, *W2;                        // bf16x6 image triples (dvs_bf16.h): fp32-accurate products on the bf16 pipe
    float *b1, *b2, *lg, *lb, *ng, *nb;
};
DVS_HD inline FfnLds ffn_lds(char* smem) {
    FfnLds l;
    l.W1 = (dvs_bf16*)smem;
    l.W2 = l.W1 + 3 * 64 * DVS_LDB;
    l.b1 = (float*)(l.W2 + 3 * 64 * DVS_LDB);
    l.b2 = l.b1 + 64;
    l.lg = l.b2 + 64;
    l.lb = l.lg + 64;
    l.ng = l.lb + 64;
    l.nb = l.ng + 64;
    return l;
}
static size_t ffn_lds_bytes() { return 6 * 64 * DVS_LDB * sizeof(dvs_bf16) + 6 * 64 * sizeof(float); }

inline void ffn_plan(DvsStagePlan& p, const FfnArgs& a, char* smem) {
    const FfnLds l = ffn_lds(smem);
    dvs_plan_clear(p);
    dvs_plan_seg(p, smem, l.W1, (const dvs_bf16*)a.wimg + DvsFfnImg::W1, (int)(6 * DVS_IMG64));     // W1, W2 bf16x6 images
    dvs_plan_vec(p, smem, l.b1, a.l1_b, 64);
    dvs_plan_vec(p, smem, l.b2, a.l2_b, 64);
    dvs_plan_vec(p, smem, l.lg, a.ln.g, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.lb, a.ln.b, a.ln.stats ? 64 : 0);
    dvs_plan_vec(p, smem, l.ng, a.ng, a.out_norm ? 64 : 0);
    dvs_plan_vec(p, smem, l.nb, a.nb, a.out_norm ? 64 : 0);
    dvs_plan_seal(p);
}

template <class PP>
__device__ __forceinline__ void dvs_ffn_fwd_phase(const FfnArgs& a, char* smem, PP mine, bool stage_mine, PP next,
                                                  bool has_next) {
    const FfnLds l = ffn_lds(smem);
    DVS_STAMP(dvs_stamps_fwd, mine, 0);
    if (stage_mine) {
        dvs_stage_now<DVS_PF_FWD>(mine, smem);
        __syncthreads();
    }
    DVS_STAMP(dvs_stamps_fwd, mine, 1);
    const Lane L = dvs_lane();
    const DvsDrop D = dvs_drop_of(a.dims);
    const int ntiles = a.dims.B * a.dims.NT;
    bool gate = !stage_mine;                 // chained: the barrier that publishes this phase's images (DVS_PHASE_GATE)
    DVS_PHASE_GATE_INIT(gate);
    dvs_stagger(L.wave);
    for (int base = dvs_bid() * L.nwaves; base < ntiles; base += gridDim.x * L.nwaves) {      // uniform round count (gate)
        const int tile = base + L.wave;
        if (tile >= ntiles) {
            DVS_PHASE_GATE(gate);
            continue;
        }
        const DvsTile T = dvs_tile_of(tile, a.dims);
        f4 x[4], dummy[4];
        float rstd;
        {
            DvsRawX rx;
            dvs_load_x_issue(rx, a.xin, a.ln, tile, L);
            DVS_PHASE_GATE(gate);              // behind the round's global loads, ahead of the first LDS access
            dvs_load_x_finish<false>(x, dummy, rstd, rx, a.ln, l.lg, l.lb, T.Nl, L);
        }
        const uint32_t gdag = a.dims.dag_offset + T.dag;
        f4 h[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = dvs_vecT(l.b1, t, L);
        dvs_matb3<4>(h, dvs_split3_T(x), l.W1, 64, 0, L);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) h[t][kk] = fmaxf(h[t][kk], 0.f);
        dvs_dropout_tile(h, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_hidden, gdag), D, L, T.tok0);
        f4 y[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) y[t] = dvs_vecT(l.b2, t, L);
        dvs_matb3<4>(y, dvs_split3_T(h), l.W2, 64, 0, L);
        dvs_dropout_tile(y, dvs_site_key(a.dims.seed_lo, a.dims.seed_hi, a.site_post, gdag), D, L, T.tok0);
        const bool valid = L.r < T.Nl;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) y[t][kk] = valid ? x[t][kk] + y[t][kk] : 0.f;
        float mean, rs;
        dvs_ln_stats(y, mean, rs);
        dvs_store_tile(a.out_pre, tile, y, L);
        if (L.g == 0) {
            a.out_stats[(size_t)tile * 32 + L.r] = mean;
            a.out_stats[(size_t)tile * 32 + 16 + L.r] = rs;
        }
        if (a.out_norm) {
            f4 xn[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f4 g = dvs_vecT(l.ng, t, L), b = dvs_vecT(l.nb, t, L);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) xn[t][kk] = valid ? (y[t][kk] - mean) * rs * g[kk] + b[kk] : 0.f;
            }
            dvs_store_tile(a.out_norm, tile, xn, L);
        }
    }
    DVS_PHASE_GATE(gate);                    // a workgroup without a tile
    DVS_STAMP(dvs_stamps_fwd, mine, 2);
    if (has_next) {
        // the older wave group fetches the next phase's images while it waits for the younger one (dvs_stage.h)
        DvsPrefetch<DVS_PF_FWD_TAIL> pf;
        const bool fetcher = dvs_tid() < DVS_PF_THREADS;
        if (fetcher) dvs_prefetch_issue<false>(pf, next, dvs_tid(), DVS_PF_THREADS);
        DVS_STAMP(dvs_stamps_fwd, mine, 3);
        dvs_lds_barrier();               // every wave is done with this phase's images
        DVS_STAMP(dvs_stamps_fwd, mine, 4);
#ifdef DVS_STAMPS_ICACHE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DVS_STAMP(dvs_stamps_fwd, mine, 5);
#endif
        if (fetcher) dvs_prefetch_commit<false>(pf, next, smem, dvs_tid(), DVS_PF_THREADS);
        DVS_STAMP(dvs_stamps_fwd, mine, 6);
#ifdef DVS_STAMPS_ICACHE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (fetcher) dvs_prefetch_commit<false>(pf, next, smem, dvs_tid(), DVS_PF_THREADS);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        DVS_STAMP(dvs_stamps_fwd, mine, 7);
#endif
    }
}

// 16 waves per workgroup (4 waves per SIMD): at B = 4096 every wave owns exactly one DAG.  nw = 4 (narrow mapping): the
// same kernel on 4 waves — DVS_PF_FWD slots per thread cover the 54 wave chunks of the FFN images from 4 waves up.
__global__ __launch_bounds__(1024) void k_ffn_fwd(FfnArgs a, DvsStagePlan plan) {
    DVS_DYN_LDS(smem);
    dvs_ffn_fwd_phase(a, smem, &plan, true, &plan, false);
}

void dvs_launch_ffn_fwd(const FfnArgs& a, int grid, int nw, dvs_stream_t st) {
    const size_t lds = ffn_lds_bytes();
    DvsStagePlan plan;
    ffn_plan(plan, a, DVS_FAKE_LDS);
    DVS_SET_LDS(k_ffn_fwd, lds);
    DVS_LAUNCH(k_ffn_fwd, dim3(grid), dim3(nw == 4 ? 256 : 1024), lds, st, a, plan);
}

// ---------------------------------------------------------------------------------------------------------
// Up to DVS_FWD_STACK_PHASES consecutive sublayers of the encoder / decoder in ONE launch (one-tile path), 8 waves per
// workgroup.  Both phase kinds map tile `blockIdx.x * 8 + wave (+ gridDim.x * 8 ...)` to the same wave, so a phase reads
// tiles and LayerNorm statistics its own workgroup wrote; a workgroup barrier is the only synchronisation (k_bwd_stack).
// ---------------------------------------------------------------------------------------------------------
template <int TAG, int NW>
__global__ __launch_bounds__(64 * NW) void k_fwd_stack(FwdStackArgs s) {
    DVS_DYN_LDS(smem);
    // the plan table is read straight from the kernel-argument segment (dvs_stage.h): the struct is the only explicit argument
#ifndef DVS_EMU
    const DvsPlanK plans =
        ((const __attribute__((address_space(4))) FwdStackArgs*)__builtin_amdgcn_kernarg_segment_ptr())->plan;
#else
    const DvsPlanK plans = s.plan;
#endif
    for (int i = 0; i < s.nphase; ++i) {
        const FwdPhase& ph = s.ph[i];
        const bool first = i == 0, more = i + 1 < s.nphase;
        const DvsPlanK mine = plans + i;
        if (TAG == 0 && NW == 8 && ph.kind == DVS_FPH_LATENT) {
            // the workgroup's DAGs, two 8-DAG runs (= two rounds of the phases' DAG loops) per group; the encoder output tiles
            // were written by other waves of this workgroup: __syncthreads waits for every wave's stores (vmcnt) first
            const int step = (int)gridDim.x * 8, B = ph.u.l.dims.B;
            for (int base = dvs_bid() * 8; base < B; base += 2 * step) {
                __syncthreads();
                dvs_latent_fwd_group(ph.u.l, smem, base, base + step);
            }
            continue;
        }
        const bool more_img = more && s.ph[i + 1].kind != DVS_FPH_LATENT;      // the latent phase stages nothing
        const DvsPlanK next = plans + (more_img ? i + 1 : i);
        if (ph.kind == DVS_FPH_ATTN) dvs_attn_fwd_phase<NW>(ph.u.a, smem, mine, first, next, more_img);
        else dvs_ffn_fwd_phase(ph.u.f, smem, mine, first, next, more_img);
        // the barrier that publishes the next phase's staged images is taken by that phase, behind its first round's global
        // loads (DVS_PHASE_GATE, dvs_kernels.h); the latent phase opens with __syncthreads
    }
}

void dvs_launch_fwd_stack(const FwdStackArgs& s_in, int tag, int grid, int nw, dvs_stream_t st) {
    FwdStackArgs s = s_in;
    for (int i = 0; i < s.nphase; ++i) {
        if (s.ph[i].kind == DVS_FPH_ATTN) attn_plan(s.plan[i], s.ph[i].u.a, DVS_FAKE_LDS);
        else if (s.ph[i].kind == DVS_FPH_FFN) ffn_plan(s.plan[i], s.ph[i].u.f, DVS_FAKE_LDS);
        else dvs_plan_clear(s.plan[i]);
        s.plan[i].phase = i;
    }
#ifdef DVS_STAMPS
    for (int i = 0; i < s.nphase; ++i) s.plan[i].phase = tag * DVS_FWD_STACK_PHASES + i;      // encoder 0..5, decoder 9..17
#endif
    const size_t la = attn_lds_bytes(), lf = ffn_lds_bytes();
    const size_t lds = la > lf ? la : lf;
#define DVS_FSTACK_LAUNCH(TG, NWV)                                                             \
    do {                                                                                       \
        DVS_SET_LDS((k_fwd_stack<TG, NWV>), lds);                                              \
        DVS_LAUNCH_AS("k_fwd_stack<" #TG ">", (k_fwd_stack<TG, NWV>), dim3(grid), dim3(64 * NWV), lds, st, s); \
    } while (0)
    if (nw == 4) {
        if (tag == 0) DVS_FSTACK_LAUNCH(0, 4);
        else DVS_FSTACK_LAUNCH(1, 4);
    } else {
        if (tag == 0) DVS_FSTACK_LAUNCH(0, 8);
        else DVS_FSTACK_LAUNCH(1, 8);
    }
}

// frag-order [tiles][1024] -> natural [tiles][16][64] (= [B][16*NT][64]) (debug / tests)
__global__ void k_unfrag(const float* frag, float* out, int B) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * 1024) return;
    const size_t dag = i >> 10;
    const int e = (int)(i & 1023);
    const int t = e >> 8, lane = (e >> 2) & 63, kk = e & 3;
    const int r = lane & 15, g = lane >> 4;
    out[dag * 1024 + r * 64 + 16 * t + 4 * g + kk] = frag[i];
}
void dvs_launch_unfrag(const float* frag, float* out, int B, dvs_stream_t st) {
    const size_t n = (size_t)B * 1024;
    DVS_LAUNCH(k_unfrag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, frag, out, B);
}

#ifdef DVS_STAMPS
extern "C" int dvs_debug_read_stamps_attn(void* out, size_t bytes, int clear) {
    if (bytes > sizeof(dvs_stamps_attn)) bytes = sizeof(dvs_stamps_attn);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dvs_stamps_attn), bytes) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dvs_stamps_attn)) != hipSuccess || hipMemset(p, 0, sizeof(dvs_stamps_attn)) != hipSuccess) return 2;
    }
    return 0;
}
extern "C" int dvs_debug_read_stamps_fwd(void* out, size_t bytes, int clear) {
    if (bytes > sizeof(dvs_stamps_fwd)) bytes = sizeof(dvs_stamps_fwd);
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(dvs_stamps_fwd), bytes) != hipSuccess) return 1;
    if (clear) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(dvs_stamps_fwd)) != hipSuccess || hipMemset(p, 0, sizeof(dvs_stamps_fwd)) != hipSuccess) return 2;
    }
    return 0;
}
#endif
